// capi.cpp -- C entry points of libfamsa_host.so.
// The host logic (tree builders, Newick / CSV writers, the working-order bookkeeping) behind a
// plain C interface: used by the famsa-gpu tool's tests to drive it from Python, and usable by a
// host program that already holds an LCS matrix.  `famsa_host_*_from_matrix` consume a
// caller-supplied oriented matrix (the CPU tests pass the oracle's); `famsa_host_*_gpu` take the
// values from the MI355X engine.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "lcs_source.h"
#include "pipeline.h"
#include "seqset.h"
#include "trees.h"

using namespace famsa_host;

namespace {
thread_local std::string g_error;

// heuristic: 0 none, 1 parttree, 2 medoidtree; the int/float params override CParams::medoid when > 0
TreeOptions make_options(const char* method, int distance, int keep_duplicates, int heuristic, int subtree_size,
                         int sample_size, int threshold, float cluster_fraction, int cluster_iters)
{
    TreeOptions o;
    o.method = gt_from_string(method);
    o.dist = (Distance)distance;
    o.keep_duplicates = keep_duplicates != 0;
    o.heuristic = heuristic;
    if (subtree_size > 0) o.fast.subtree_size = subtree_size;
    if (sample_size > 0) o.fast.sample_size = sample_size;
    if (threshold > 0) o.fast.threshold = threshold;
    if (cluster_fraction > 0) o.fast.cluster_fraction = cluster_fraction;
    if (cluster_iters > 0) o.fast.cluster_iters = cluster_iters;
    o.fast.n_threads = std::max(1, host_test_int("threads", o.fast.n_threads));
    return o;
}
int fail(const std::exception& e)
{
    g_error = e.what();
    return -1;
}
// -1 is the error code of every entry point; a result that does not fit returns -(bytes needed) - 1 <= -2
// (bytes needed = size + 1 for the terminator) and says so in the error text
long give(const std::string& s, char* out, long cap)
{
    if ((long)s.size() + 1 > cap) {
        g_error = "buffer too small: need " + std::to_string(s.size() + 1) + " bytes";
        return -(long)(s.size() + 1) - 1;
    }
    memcpy(out, s.c_str(), s.size() + 1);
    return (long)s.size();
}
} // namespace

extern "C" {

const char* famsa_host_last_error(void) { return g_error.c_str(); }

// FAMSA's working order of a FASTA file: sorted2input[n], sorted2unique[n]; returns n_unique.
int famsa_host_workset(const char* fasta, int keep_duplicates, int* sorted2input, int* sorted2unique, int cap)
{
    try {
        SeqSet s = load_fasta(fasta);
        WorkSet w = make_workset(s, keep_duplicates != 0);
        if ((int)s.size() > cap) throw std::runtime_error("buffer too small");
        for (int i = 0; i < w.n_sorted(); ++i) {
            sorted2input[i] = w.sorted2input[i];
            sorted2unique[i] = w.sorted2unique[i];
        }
        return w.n_unique();
    } catch (const std::exception& e) {
        return fail(e);
    }
}

// Newick text of `-gt <method> -gt_export` for `fasta`, LCS values taken from `square`
// (input order, square[ref*n + partner], n = number of FASTA records).
long famsa_host_tree_from_matrix(const char* fasta, const uint32_t* square, const char* method, int distance,
                                 int keep_duplicates, int heuristic, int subtree_size, int sample_size,
                                 int threshold, float cluster_fraction, int cluster_iters, char* out, long cap)
{
    try {
        SeqSet s = load_fasta(fasta);
        return give(guide_tree_newick_from_matrix(s, square,
                                                   make_options(method, distance, keep_duplicates, heuristic, subtree_size,
                                                                sample_size, threshold, cluster_fraction, cluster_iters)),
                    out, cap);
    } catch (const std::exception& e) {
        return fail(e);
    }
}

// The same with the rest of the CLI's tree options: -num_evals (0 = default), -gt chained's seed, -dump_seeds (NULL = none).
long famsa_host_tree_from_matrix_ex(const char* fasta, const uint32_t* square, const char* method, int distance,
                                    int keep_duplicates, int heuristic, int subtree_size, int sample_size,
                                    int threshold, float cluster_fraction, int cluster_iters, int num_evals,
                                    uint32_t chained_seed, const char* dump_seeds_path, char* out, long cap)
{
    try {
        SeqSet s = load_fasta(fasta);
        TreeOptions o = make_options(method, distance, keep_duplicates, heuristic, subtree_size, sample_size, threshold,
                                     cluster_fraction, cluster_iters);
        if (num_evals > 0) o.fast.num_evaluations = num_evals;
        o.chained_seed = chained_seed;
        if (dump_seeds_path) o.dump_seeds_path = dump_seeds_path;
        return give(guide_tree_newick_from_matrix(s, square, o), out, cap);
    } catch (const std::exception& e) {
        return fail(e);
    }
}

// `-dist_export [-pid] [-square_matrix]` written to `csv_path`, LCS values from `square`.
int famsa_host_dist_export_from_matrix(const char* fasta, const uint32_t* square, int distance, int square_matrix,
                                       int pid, const char* csv_path)
{
    try {
        SeqSet s = load_fasta(fasta);
        const int n = (int)s.size();
        std::vector<uint32_t> lens(n);
        for (int i = 0; i < n; ++i) lens[i] = s.length(i);
        MatrixLcsSource src(n, lens.data(), square);
        write_distance_csv(src, s.ids, (Distance)distance, square_matrix != 0, pid != 0, csv_path);
        return 0;
    } catch (const std::exception& e) {
        return fail(e);
    }
}

// The same two operations with the LCS computed on GPU `device`.
long famsa_host_tree_gpu(const char* fasta, int device, const char* method, int distance, int keep_duplicates,
                         int heuristic, int subtree_size, int sample_size, int threshold, float cluster_fraction,
                         int cluster_iters, char* out, long cap)
{
    try {
        SeqSet s = load_fasta(fasta);
        return give(guide_tree_newick_gpu(s, device,
                                           make_options(method, distance, keep_duplicates, heuristic, subtree_size,
                                                        sample_size, threshold, cluster_fraction, cluster_iters),
                                           nullptr),
                    out, cap);
    } catch (const std::exception& e) {
        return fail(e);
    }
}

int famsa_host_dist_export_gpu(const char* fasta, int device, int distance, int square_matrix, int pid,
                               const char* csv_path)
{
    try {
        SeqSet s = load_fasta(fasta);
        dist_export_gpu(s, device, (Distance)distance, square_matrix != 0, pid != 0, csv_path, nullptr);
        return 0;
    } catch (const std::exception& e) {
        return fail(e);
    }
}

// The records of a FASTA file as the reader delivers them (tests): ids joined by '\n' into ids_buf,
// codes and offsets as in SeqSet.  Returns the record count, or -1 (error / buffers too small).
long famsa_host_records(const char* fasta, int n_threads, char* ids_buf, long ids_cap, uint8_t* codes_buf,
                        long codes_cap, uint64_t* offsets, long cap)
{
    try {
        SeqSet s = load_fasta(fasta, n_threads);
        std::string joined;
        for (const auto& id : s.ids) {
            joined += id;
            joined += '\n';
        }
        if ((long)s.size() + 1 > cap || (long)s.codes.size() > codes_cap || (long)joined.size() + 1 > ids_cap)
            throw std::runtime_error("buffer too small");
        memcpy(ids_buf, joined.c_str(), joined.size() + 1);
        if (!s.codes.empty()) memcpy(codes_buf, s.codes.data(), s.codes.size());
        for (size_t i = 0; i <= s.size(); ++i) offsets[i] = s.offsets[i];
        return (long)s.size();
    } catch (const std::exception& e) {
        return fail(e);
    }
}

int famsa_host_format_distance(double v, char* out) { return format_distance(v, out); }

// Newick text of a tree given as child arrays (node i < n_leaves: a leaf named names[i]; node i >= n_leaves: children
// left[i - n_leaves], right[i - n_leaves]; the last node is the root), after GuideTree::fromUnique over `sorted2unique`
// when that is not NULL (then the arrays describe the tree of the unique sequences and names has one entry per record).
long famsa_host_newick(const int32_t* left, const int32_t* right, int n_leaves, int n_internal, const char* const* names,
                       int n_names, const int32_t* sorted2unique, char* out, long cap)
{
    try {
        if (n_leaves < 0 || n_internal < 0 || n_names < 0 || (n_leaves + n_internal > 0 && (!names || (n_internal > 0 && (!left || !right)))))
            throw std::runtime_error("famsa_host_newick: bad argument");
        if (sorted2unique ? n_names < n_leaves : n_names != n_leaves)
            throw std::runtime_error("famsa_host_newick: " + std::to_string(n_names) + " names for " + std::to_string(n_leaves) + " leaves");
        std::vector<char> seen((size_t)n_leaves + n_internal, 0);
        for (int i = 0; i < n_internal; ++i)
            for (int c : {left[i], right[i]}) { // a child is an earlier or later node of the tree, and of one parent only
                if (c < 0 || c >= n_leaves + n_internal || c == n_leaves + i || seen[c])
                    throw std::runtime_error("famsa_host_newick: node " + std::to_string(n_leaves + i) + " has the child " + std::to_string(c) +
                                             (c >= 0 && c < n_leaves + n_internal && seen[c] ? ", which has a parent already" : ", which is not a node of the tree"));
                seen[c] = 1;
            }
        tree_structure tree((size_t)n_leaves, node_t(-1, -1));
        for (int i = 0; i < n_internal; ++i) tree.emplace_back(left[i], right[i]);
        if (sorted2unique) tree_from_unique(tree, std::vector<int>(sorted2unique, sorted2unique + n_names));
        return give(tree_to_newick(tree, std::vector<const char*>(names, names + n_names)), out, cap);
    } catch (const std::exception& e) {
        return fail(e);
    }
}

// The host CLARANS search over a caller-supplied float distance triangle (tests compare the device search with it).
int famsa_host_clarans(const float* triangle, int n_elems, int n_medoids, int n_fixed, float explore_fraction,
                       int num_local, int* medoids)
{
    try {
        clarans_host(triangle, n_elems, n_medoids, n_fixed, explore_fraction, num_local, medoids);
        return 0;
    } catch (const std::exception& e) {
        return fail(e);
    }
}

} // extern "C"
