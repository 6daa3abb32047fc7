// ref_harness.cpp -- thin driver over the REFERENCE's own hot-path sources.
//
// TEST INFRASTRUCTURE ONLY (same rule as lcs_oracle.c): only tests/, smoke() and
// bench.py's cpu_baseline leg may load the library this builds.
//
// This file is ours; it #includes the reference's headers and is linked by
// oracle/Makefile against object files compiled from the reference sources where
// they lie under /root/reference (nothing is copied into the repo, outputs go to
// oracle/_ref/ which is git-ignored).  The full `famsa` binary is NOT buildable
// in this image (src/core/io_service.h needs the empty libs/{libdeflate,zlib-ng,
// isa-l} submodules), so the orchestration steps of CFAMSA::ComputeMSA that sit
// around the guide-tree generators (msa.cpp:245-356, 518-584) are driven from
// here; that this driving is faithful is pinned by the reference's own golden
// files (tests/test_ref_goldens.py: adeno_fiber sl/slink/upgma .dnd,
// dist/pid .csv, hemopexin medoid-*.dnd all reproduced byte for byte).
#include "core/sequence.h"
#include "lcs/lcsbp.h"
#include "tree/AbstractTreeGenerator.hpp"
#include "tree/DistanceCalculator.h"
#include "tree/FastTree.h"
#include "tree/GuideTree.h"
#include "tree/MSTPrim.h"
#include "tree/NeighborJoining.h"
#include "tree/NewickParser.h"
#include "tree/SingleLinkage.h"
#include "tree/UPGMA.h"

#include <algorithm>
#include <chrono>
#include <cstring>
#include <fstream>
#include <memory>
#include <numeric>
#include <string>
#include <vector>

namespace {

// Called with the working set right before a generator runs -- the place of AbstractTreeGenerator::operator()
// (reference tree/AbstractTreeGenerator.cpp:25-32).  Unset in libfamsa_ref.so (the CPU reference); set by
// gpu_lcsbp.cpp in libfamsa_gpuref.so, whose CLCSBP is served by the GPU engine and needs the set uploaded.
void (*g_pre_run)(CSequence* const* seqs, int n, void* user) = nullptr;
void* g_pre_run_user = nullptr;
void pre_run(std::vector<CSequence*>& working_set)
{
    if (g_pre_run) g_pre_run(working_set.data(), (int)working_set.size(), g_pre_run_user);
}

struct RefSet {
    std::vector<std::string> ids, residues; // input order, as read
    std::vector<CSequence> seqs;            // encoded once by the reference's CSequence ctor
};

instruction_set_t isa_of(int isa)
{
    switch (isa) {
    case 0: return instruction_set_t::none;
    case 1: return instruction_set_t::avx;
    default: return instruction_set_t::avx2;
    }
}

// FASTA reader with the line handling of IOService::loadFasta (io_service.h:84-127).
void read_fasta(const std::string& path, RefSet& rs)
{
    std::ifstream f(path, std::ios::binary);
    if (!f.good())
        throw std::runtime_error("cannot open " + path);
    std::string s, id, seq;
    int no = 0;
    while (std::getline(f, s)) {
        while (!s.empty() && (s.back() == '\n' || s.back() == '\r'))
            s.pop_back();
        if (s.empty())
            continue;
        if (s[0] == '>') {
            if (!id.empty() && !seq.empty()) {
                rs.ids.push_back(id);
                rs.residues.push_back(seq);
                rs.seqs.emplace_back(id, seq, no++, nullptr);
                seq.clear();
            }
            id = s;
        } else
            seq += s;
    }
    if (!id.empty() && !seq.empty()) {
        rs.ids.push_back(id);
        rs.residues.push_back(seq);
        rs.seqs.emplace_back(id, seq, no++, nullptr);
    }
}

// Work on a private deep copy so a handle can be reused.
std::vector<CSequence> clone(const RefSet& rs)
{
    std::vector<CSequence> v;
    v.reserve(rs.ids.size());
    for (size_t i = 0; i < rs.ids.size(); ++i)
        v.emplace_back(rs.ids[i], rs.residues[i], (int)i, nullptr);
    return v;
}

// Equivalent of CFAMSA::sortAndExtendSequences (msa.cpp:245-279).
void sort_and_extend(std::vector<CSequence>& sequences)
{
    std::vector<CSequence*> ptrs(sequences.size());
    std::transform(sequences.begin(), sequences.end(), ptrs.begin(), [](CSequence& s) { return &s; });
    std::stable_sort(ptrs.begin(), ptrs.end(), [](const CSequence* a, const CSequence* b) {
        return a->length > b->length ||
               (a->length == b->length &&
                std::lexicographical_compare(a->data, a->data + a->data_size, b->data, b->data + b->data_size));
    });
    uint32_t max_len = ptrs[0]->length;
    std::vector<int> order(ptrs.size());
    for (size_t i = 0; i < ptrs.size(); ++i)
        order[i] = (int)(ptrs[i] - sequences.data());
    std::vector<CSequence> output;
    output.reserve(sequences.size());
    for (size_t i = 0; i < order.size(); ++i) {
        output.emplace_back(std::move(sequences[order[i]]));
        output.back().DataResize(max_len, UNKNOWN_SYMBOL);
    }
    output.swap(sequences);
}

// CFAMSA::extendSequences (msa.cpp:282-317).
void extend(std::vector<CSequence>& sequences)
{
    uint32_t max_len = 0;
    for (auto& s : sequences)
        max_len = std::max(max_len, s.length);
    for (auto& s : sequences)
        s.DataResize(max_len, UNKNOWN_SYMBOL);
}

// CFAMSA::removeDuplicates (msa.cpp:338-356).
void remove_duplicates(std::vector<CSequence*>& sorted, std::vector<int>& original2sorted)
{
    auto eq = [](const CSequence* a, const CSequence* b) {
        return a->length == b->length && std::equal(a->data, a->data + a->length, b->data);
    };
    int cur = 0;
    for (int i = 1; i < (int)sorted.size(); ++i) {
        if (!eq(sorted[i], sorted[i - 1]))
            ++cur;
        original2sorted[i] = cur;
    }
    sorted.erase(std::unique(sorted.begin(), sorted.end(), eq), sorted.end());
}

template <Distance D>
std::shared_ptr<AbstractTreeGenerator> make_gen(int gt, bool heuristic_on, int n_threads, instruction_set_t isa)
{
    // CFAMSA::createTreeGenerator, msa.cpp:134-169.  gt: 0 sl(MST_Prim) 1 slink 2 upgma 3 nj 4 upgma_modified
    if (gt == 1 || (heuristic_on && gt == 0))
        return std::make_shared<SingleLinkage<D>>(n_threads, isa);
    if (gt == 0)
        return std::make_shared<MSTPrim<D>>(n_threads, isa);
    if (gt == 2 || gt == 4)
        return std::make_shared<UPGMA<D>>(n_threads, isa, gt == 4);
    if (gt == 3)
        return std::make_shared<NeighborJoining<D>>(n_threads, isa);
    throw std::runtime_error("bad gt");
}

struct MedoidParams { // CParams::medoid, core/params.h:88-97
    int subtree_size = 100, sample_size = 2000, num_evaluations = 1, threshold = 2000;
    float cluster_fraction = 0.1f;
    int cluster_iters = 2;
    std::string seed_file_name; // CParams::seed_file_name (-dump_seeds)
};

// the observer createTreeGenerator registers for -dump_seeds (msa.cpp:184-199), restated: the depth-0 seeds' ids
class SeedDumper : public IFastTreeObserver {
    std::ofstream ofs;
public:
    explicit SeedDumper(const std::string& fname) { ofs.open(fname); }
    void notifySeedsSelected(const std::vector<CSequence*>& seeds, int depth) override
    {
        if (depth == 0)
            for (const auto s : seeds) ofs << s->id.substr(1) << std::endl;
    }
};

template <Distance D>
std::string run_tree(const RefSet& rs, int gt, int heuristic, const MedoidParams& mp, bool keep_dups,
                     int n_threads, instruction_set_t isa)
{
    std::vector<CSequence> sequences = clone(rs);
    // CFAMSA::adjustParams, msa.cpp:83-88
    if (heuristic != 0 && (int)sequences.size() < mp.threshold)
        heuristic = 0;

    GuideTree tree;
    sort_and_extend(sequences);
    std::vector<CSequence*> mapped(sequences.size());
    std::transform(sequences.begin(), sequences.end(), mapped.begin(), [](CSequence& s) { return &s; });
    std::vector<int> original2mapped(sequences.size());
    std::iota(original2mapped.begin(), original2mapped.end(), 0);
    if (!keep_dups)
        remove_duplicates(mapped, original2mapped);
    if (mapped.size() == 1)
        return std::string();
    for (int i = 0; i < (int)mapped.size(); ++i)
        mapped[i]->sequence_no = i;

    std::shared_ptr<AbstractTreeGenerator> gen = make_gen<D>(gt, heuristic != 0, n_threads, isa);
    if (heuristic != 0) { // msa.cpp:172-239; heuristic 1 = parttree, 2 = medoidtree
        std::shared_ptr<IClustering> clustering =
            (heuristic == 1) ? nullptr : std::make_shared<CLARANS>(mp.cluster_fraction, mp.cluster_iters);
        auto ft = std::make_shared<FastTree<D>>(n_threads, isa, std::dynamic_pointer_cast<IPartialGenerator>(gen),
                                                mp.subtree_size, mp.sample_size, mp.num_evaluations, mp.threshold,
                                                clustering);
        if (!mp.seed_file_name.empty()) ft->registerObserver(std::make_shared<SeedDumper>(mp.seed_file_name));
        gen = ft;
    }
    pre_run(mapped);
    (*gen)(mapped, tree.raw());
    for (auto& s : sequences) // shrinkSequences, msa.cpp:320-335
        s.DataResize(s.length, UNKNOWN_SYMBOL);
    tree.fromUnique(original2mapped);
    std::string description;
    NewickParser nw(false);
    nw.store(sequences, tree.raw(), description);
    return description;
}

} // namespace

extern "C" {

void ref_set_pre_run_hook(void (*hook)(CSequence* const* seqs, int n, void* user), void* user)
{
    g_pre_run = hook;
    g_pre_run_user = user;
}

void* ref_open_fasta(const char* path)
{
    try {
        auto* rs = new RefSet;
        read_fasta(path, *rs);
        return rs;
    } catch (...) {
        return nullptr;
    }
}

void* ref_open_seqs(const char* const* ids, const char* const* residues, int n)
{
    auto* rs = new RefSet;
    rs->seqs.reserve(n);
    for (int i = 0; i < n; ++i) {
        rs->ids.emplace_back(ids[i]);
        rs->residues.emplace_back(residues[i]);
        rs->seqs.emplace_back(rs->ids.back(), rs->residues.back(), i, nullptr);
    }
    return rs;
}

void ref_close(void* h) { delete (RefSet*)h; }
int ref_count(void* h) { return (int)((RefSet*)h)->seqs.size(); }
int ref_length(void* h, int i) { return (int)((RefSet*)h)->seqs[i].length; }
// symbol codes of sequence i as the reference's own encoder produced them
void ref_codes(void* h, int i, unsigned char* out)
{
    auto& s = ((RefSet*)h)->seqs[i];
    for (uint32_t k = 0; k < s.length; ++k)
        out[k] = (unsigned char)s.data[k];
}

// out[r*n_cols+c] = LCS(ref = ref_ids[r], partner = col_ids[c]) through the reference's
// dispatcher CLCSBP::GetLCSBP (lcs/lcsbp.cpp:163), fed in groups of 8 like
// calculateDistanceVector (AbstractTreeGenerator.hpp:130-182): a trailing partial group is
// padded with nullptr and therefore takes the classic path.  Input order, padded sequences.
int ref_lcs_rect(void* h, const int* ref_ids, int n_refs, const int* col_ids, int n_cols, int isa, uint32_t* out)
{
    std::vector<CSequence> seqs = clone(*(RefSet*)h);
    extend(seqs);
    for (int i = 0; i < (int)seqs.size(); ++i)
        seqs[i].sequence_no = i;
    {
        std::vector<CSequence*> all(seqs.size());
        std::transform(seqs.begin(), seqs.end(), all.begin(), [](CSequence& s) { return &s; });
        pre_run(all);
    }
    CLCSBP lcsbp(isa_of(isa));
    uint32_t lens[8];
    for (int r = 0; r < n_refs; ++r) {
        CSequence* ref = &seqs[ref_ids[r]];
        ref->ComputeBitMasks();
        for (int c0 = 0; c0 < n_cols; c0 += 8) {
            CSequence* p[8];
            for (int k = 0; k < 8; ++k)
                p[k] = (c0 + k < n_cols) ? &seqs[col_ids[c0 + k]] : nullptr;
            lcsbp.GetLCSBP(ref, p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], lens);
            for (int k = 0; k < 8 && c0 + k < n_cols; ++k)
                out[(size_t)r * n_cols + c0 + k] = lens[k];
        }
        ref->ReleaseBitMasks();
    }
    return 0;
}

// Newick text of `famsa -gt <gt> [-medoidtree|-parttree] [-keep-duplicates] -gt_export`.
// distance: 0 indel_div_lcs, 1 indel075_div_lcs (CLI default). Returns length, or -1.
long ref_tree_newick(void* h, int gt, int distance, int heuristic, int subtree_size, int sample_size,
                     int threshold, float cluster_fraction, int cluster_iters, int keep_dups, int n_threads,
                     int isa, char* out, long cap)
{
    try {
        MedoidParams mp;
        if (subtree_size > 0) mp.subtree_size = subtree_size;
        if (sample_size > 0) mp.sample_size = sample_size;
        if (threshold > 0) mp.threshold = threshold;
        if (cluster_fraction > 0) mp.cluster_fraction = cluster_fraction;
        if (cluster_iters > 0) mp.cluster_iters = cluster_iters;
        std::string s = (distance == 0)
            ? run_tree<Distance::indel_div_lcs>(*(RefSet*)h, gt, heuristic, mp, keep_dups != 0, n_threads, isa_of(isa))
            : run_tree<Distance::indel075_div_lcs>(*(RefSet*)h, gt, heuristic, mp, keep_dups != 0, n_threads, isa_of(isa));
        if ((long)s.size() + 1 > cap)
            return -(long)s.size() - 1;
        memcpy(out, s.c_str(), s.size() + 1);
        return (long)s.size();
    } catch (...) {
        return -1;
    }
}

// ref_tree_newick plus the two parameters it does not carry: -num_evals (core/params.cpp:206) and -dump_seeds
// (params.cpp:217; seeds_path NULL or "" = none).
long ref_tree_newick_ex(void* h, int gt, int distance, int heuristic, int subtree_size, int sample_size,
                        int threshold, float cluster_fraction, int cluster_iters, int num_evals, const char* seeds_path,
                        int keep_dups, int n_threads, int isa, char* out, long cap)
{
    try {
        MedoidParams mp;
        if (subtree_size > 0) mp.subtree_size = subtree_size;
        if (sample_size > 0) mp.sample_size = sample_size;
        if (threshold > 0) mp.threshold = threshold;
        if (cluster_fraction > 0) mp.cluster_fraction = cluster_fraction;
        if (cluster_iters > 0) mp.cluster_iters = cluster_iters;
        if (num_evals > 0) mp.num_evaluations = num_evals;
        if (seeds_path) mp.seed_file_name = seeds_path;
        std::string s = (distance == 0)
            ? run_tree<Distance::indel_div_lcs>(*(RefSet*)h, gt, heuristic, mp, keep_dups != 0, n_threads, isa_of(isa))
            : run_tree<Distance::indel075_div_lcs>(*(RefSet*)h, gt, heuristic, mp, keep_dups != 0, n_threads, isa_of(isa));
        if ((long)s.size() + 1 > cap)
            return -(long)s.size() - 1;
        memcpy(out, s.c_str(), s.size() + 1);
        return (long)s.size();
    } catch (...) {
        return -1;
    }
}

// The same orchestration with a generator the CALLER builds (oracle/adapter_check.cpp: the reference's
// generators with their distance stage bound to the GPU engine).  Returns length, -(needed) - 1, or -1.
long ref_tree_newick_with(void* h, int keep_dups, AbstractTreeGenerator* (*make)(void* user), void* user, char* out, long cap)
{
    try {
        const RefSet& rs = *(RefSet*)h;
        std::vector<CSequence> sequences = clone(rs);
        GuideTree tree;
        sort_and_extend(sequences);
        std::vector<CSequence*> mapped(sequences.size());
        std::transform(sequences.begin(), sequences.end(), mapped.begin(), [](CSequence& s) { return &s; });
        std::vector<int> original2mapped(sequences.size());
        std::iota(original2mapped.begin(), original2mapped.end(), 0);
        if (!keep_dups)
            remove_duplicates(mapped, original2mapped);
        std::string description;
        if (mapped.size() > 1) {
            for (int i = 0; i < (int)mapped.size(); ++i)
                mapped[i]->sequence_no = i;
            std::unique_ptr<AbstractTreeGenerator> gen(make(user));
            pre_run(mapped);
            (*gen)(mapped, tree.raw());
            for (auto& s : sequences)
                s.DataResize(s.length, UNKNOWN_SYMBOL);
            tree.fromUnique(original2mapped);
            NewickParser nw(false);
            nw.store(sequences, tree.raw(), description);
        }
        if ((long)description.size() + 1 > cap)
            return -(long)description.size() - 2;
        memcpy(out, description.c_str(), description.size() + 1);
        return (long)description.size();
    } catch (...) {
        return -1;
    }
}

// `famsa -dist_export [-pid] [-square_matrix]`: ComputeMSA's early branch, msa.cpp:518-526.
int ref_dist_export(void* h, int distance, int square, int pid, int n_threads, int isa, const char* out_path)
{
    try {
        std::vector<CSequence> seqs = clone(*(RefSet*)h);
        std::vector<CSequence*> mapped(seqs.size());
        std::transform(seqs.begin(), seqs.end(), mapped.begin(), [](CSequence& s) { return &s; });
        extend(seqs);
        tree_structure tree;
        std::shared_ptr<AbstractTreeGenerator> gen;
        if (distance == 0)
            gen = std::make_shared<DistanceCalculator<Distance::indel_div_lcs>>(n_threads, isa_of(isa), out_path, square != 0, pid != 0);
        else
            gen = std::make_shared<DistanceCalculator<Distance::indel075_div_lcs>>(n_threads, isa_of(isa), out_path, square != 0, pid != 0);
        for (int i = 0; i < (int)mapped.size(); ++i)
            mapped[i]->sequence_no = i; // input order (the CSequence constructor already numbered them so)
        pre_run(mapped);
        (*gen)(mapped, tree);
        return 0;
    } catch (...) {
        return -1;
    }
}

// CPU baseline: the reference's own multi-threaded all-pairs loop,
// UPGMA::computeDistances (tree/UPGMA.cpp:75-109) on the first n_use sequences in sorted order
// (sort as ComputeMSA does).  Returns seconds; *pairs and *cells (sum len_ref*len_partner) filled.
// If out_matrix != nullptr it receives the float triangle (n_use*(n_use-1)/2).
double ref_time_triangle(void* h, int n_use, int n_threads, int isa, double* pairs, double* cells, float* out_matrix)
{
    std::vector<CSequence> seqs = clone(*(RefSet*)h);
    sort_and_extend(seqs);
    if (n_use <= 0 || n_use > (int)seqs.size())
        n_use = (int)seqs.size();
    std::vector<CSequence*> mapped(n_use);
    for (int i = 0; i < n_use; ++i) {
        mapped[i] = &seqs[i];
        mapped[i]->sequence_no = i;
    }
    size_t m = (size_t)n_use * (n_use - 1) / 2;
    std::vector<float> local;
    float* mat = out_matrix;
    if (!mat) {
        local.resize(m);
        mat = local.data();
    }
    double c = 0, sum_prev = 0;
    for (int i = 0; i < n_use; ++i) {
        c += (double)seqs[i].length * sum_prev;
        sum_prev += seqs[i].length;
    }
    UPGMA<Distance::indel075_div_lcs> gen(n_threads, isa_of(isa), false);
    auto t0 = std::chrono::steady_clock::now();
    gen.computeDistances(mapped, mat);
    auto t1 = std::chrono::steady_clock::now();
    if (pairs) *pairs = (double)m;
    if (cells) *cells = c;
    return std::chrono::duration<double>(t1 - t0).count();
}

// The reference's CLARANS (tree/Clustering.cpp) over a caller-supplied float triangle.
void ref_clarans(const float* triangle, int n_elems, int n_medoids, int n_fixed, float explore_fraction, int num_local,
                 int* medoids)
{
    CLARANS search(explore_fraction, num_local);
    search(triangle, n_elems, n_medoids, n_fixed, medoids);
}

} // extern "C"
