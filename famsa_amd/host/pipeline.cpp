#include <sys/mman.h>
#include <sys/resource.h>
#include "pipeline.h"

#include <thread>
#include <unistd.h>
#include <cstdio>

#include <chrono>
#include <cstdlib>
#include <stdexcept>

namespace famsa_host {

long resident_kb()
{
    long kb = 0;
    if (FILE* f = fopen("/proc/self/statm", "r")) {
        long size = 0, res = 0;
        if (fscanf(f, "%ld %ld", &size, &res) == 2) kb = res * (sysconf(_SC_PAGESIZE) / 1024);
        fclose(f);
    }
    return kb;
}

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

std::string guide_tree_newick(const SeqSet& s, const WorkSet& w, LcsSource& src, const TreeOptions& opt, Timings* t, tree_structure* spare)
{
    const double t0 = now_s();
    if (w.n_unique() == 1) return std::string(); // the reference skips the tree stage entirely (msa.cpp:549-556)
    tree_structure tree;
    if (spare) tree.swap(*spare);
    // CFAMSA::adjustParams (msa.cpp:83-88): the heuristic is dropped for inputs below the threshold
    // (counted on ALL input records); createTreeGenerator (msa.cpp:134-239) wraps the partial generator
    int heuristic = opt.heuristic;
    if (heuristic != 0 && (int)s.size() < opt.fast.threshold) heuristic = 0;
    if (t) {
        t->n_records = (int)s.size();
        t->n_duplicates = w.n_sorted() - w.n_unique();
    }
    if (heuristic != 0) {
        FastTreeParams fp = opt.fast;
        fp.use_clustering = heuristic == 2;
        std::vector<int> top_seeds;
        if (!opt.dump_seeds_path.empty()) fp.top_seeds = &top_seeds;
        build_tree_fast(src, opt.method, opt.dist, fp, tree);
        if (!opt.dump_seeds_path.empty()) { // SeedDumper (msa.cpp:184-199): the ids without their '>', one per line
            FILE* f = fopen(opt.dump_seeds_path.c_str(), "w");
            if (!f) throw std::runtime_error("cannot open " + opt.dump_seeds_path);
            for (int u : top_seeds) {
                const std::string& id = s.ids[w.sorted2input[w.unique2sorted[u]]];
                fprintf(f, "%s\n", id.c_str() + (!id.empty() && id[0] == '>' ? 1 : 0));
            }
            fclose(f);
        }
    } else if (opt.method == GT::chained) {
        build_tree_chained(w.n_unique(), opt.chained_seed, tree);
    } else {
        build_tree(src, opt.method, opt.dist, tree, 1);
    }
    const double t1 = now_s();
    const long rss_tree = t ? resident_kb() : 0;
    // names of the leaves = ids in sorted order (the reference reorders its sequence vector): the Newick's business, not the
    // tree's -- 3 x 10^6 scattered reads, which used to sit in front of the tree stage and inside its timer
    std::vector<const char*> names(w.n_sorted());
    {
        const int n_names = w.n_sorted(), n_workers = n_names >= 200000 ? std::max(1, std::min(8, opt.fast.n_threads)) : 1;
        auto fill = [&](int k0, int k1) {
            for (int k = k0; k < k1; ++k) names[k] = s.ids[w.sorted2input[k]].c_str();
        };
        std::vector<std::thread> helpers;
        for (int t = 1; t < n_workers; ++t)
            helpers.emplace_back(fill, (int)((int64_t)n_names * t / n_workers), (int)((int64_t)n_names * (t + 1) / n_workers));
        fill(0, (int)((int64_t)n_names / n_workers));
        for (auto& h : helpers) h.join();
    }
    if (profile_on()) fprintf(stderr, "newick: the leaves' names %.3f s\n", now_s() - t1);
    tree_from_unique(tree, w.sorted2unique);
    std::string nwk = tree_to_newick(tree, names);
    if (t) {
        t->tree_s = t1 - t0;
        t->newick_s = now_s() - t1;
        t->rss_tree_kb = rss_tree;
        t->rss_newick_kb = resident_kb();
    }
    return nwk;
}

std::string guide_tree_newick_from_matrix(const SeqSet& s, const uint32_t* sq, const TreeOptions& opt)
{
    const int n_in = (int)s.size();
    WorkSet w = make_workset(s, opt.keep_duplicates);
    const int u = w.n_unique();
    std::vector<uint32_t> lens(u), m((size_t)u * u);
    std::vector<int> in_of(u);
    for (int a = 0; a < u; ++a) {
        in_of[a] = w.sorted2input[w.unique2sorted[a]];
        lens[a] = s.length(in_of[a]);
    }
    for (int a = 0; a < u; ++a)
        for (int b = 0; b < u; ++b) m[(size_t)a * u + b] = sq[(size_t)in_of[a] * n_in + in_of[b]];
    MatrixLcsSource src(u, lens.data(), m.data());
    return guide_tree_newick(s, w, src, opt);
}

bool g_abandon_engine_at_return = false;

EngineFuture start_engine(int device) { return start_engine(std::vector<int>{device}); }

EngineFuture start_engine(const std::vector<int>& devices, int expect_threads)
{
    return std::async(std::launch::async, [devices, expect_threads] {
        auto src = std::make_unique<GpuLcsSource>(devices);
        if (expect_threads > 1) src->expect_threads(expect_threads);
        return src;
    });
}

namespace {
// a large buffer goes back to the system off the calling thread
void release_in_background(Bytes& bytes)
{
    if (bytes.capacity() < (size_t)64 << 20) {
        Bytes().swap(bytes);
        return;
    }
    std::thread([held = std::move(bytes)]() mutable {
        // The pages go first, a piece at a time (MADV_DONTNEED takes the address space's lock shared, like a page fault); freeing
        // the block then unmaps an empty range.  Unmapping 765 MB (3 x 10^6 sequences) in one go held that lock exclusively for
        // 0.1 s, and the tree stage's first allocations on the main thread -- the empty tree: 48 MB -- stood still behind it.
        const uintptr_t page = 4096, piece = (uintptr_t)32 << 20;
        uintptr_t a = ((uintptr_t)held.data() + page - 1) & ~(page - 1);
        const uintptr_t end = ((uintptr_t)held.data() + held.size()) & ~(page - 1);
        for (; a < end; a += piece) (void)madvise((void*)a, (size_t)std::min(piece, end - a), MADV_DONTNEED);
        Bytes().swap(held);
    }).detach();
    Bytes().swap(bytes);
}

std::string newick_gpu(const SeqSet& s, SeqSet* consumable, int device, const TreeOptions& opt, Timings* t, EngineFuture* engine);
} // namespace

std::string guide_tree_newick_gpu(const SeqSet& s, int device, const TreeOptions& opt, Timings* t, EngineFuture* engine)
{
    return newick_gpu(s, nullptr, device, opt, t, engine);
}

std::string guide_tree_newick_gpu_consuming(SeqSet& s, int device, const TreeOptions& opt, Timings* t, EngineFuture* engine)
{
    return newick_gpu(s, &s, device, opt, t, engine);
}

namespace {
std::string newick_gpu(const SeqSet& s, SeqSet* consumable, int device, const TreeOptions& opt, Timings* t, EngineFuture* engine)
{
    EngineFuture own;
    if (!engine) {
        own = start_engine(device);
        engine = &own;
    }
    double t0 = now_s();
    WorkSet w = make_workset(s, opt.keep_duplicates, opt.fast.n_threads);
    std::vector<int> in_of(w.n_unique());
    for (int a = 0; a < w.n_unique(); ++a) in_of[a] = w.sorted2input[w.unique2sorted[a]];
    double t1 = now_s();
    // The heuristics' tree has 2n - 1 nodes (48 MB at 3 x 10^6 sequences): its pages are touched now, by a thread of its own,
    // while this one waits for the engine and the upload -- in front of the tree stage they cost 0.04 s, 0.11 s beside the
    // release of the residues below (both work on the address space: profiles/c5_stage_r06.txt).
    tree_structure spare;
    std::future<void> spare_ready;
    const bool fast_tree = opt.heuristic != 0 && (int)s.size() >= opt.fast.threshold && w.n_unique() >= 2;
    if (fast_tree && w.n_unique() >= 100000 && !host_test("no_spare_tree"))
        spare_ready = std::async(std::launch::async, [&spare, n = (size_t)w.n_unique()] { spare.assign(2 * n - 1, node_t(-1, -1)); });
    std::unique_ptr<GpuLcsSource> held = engine->get(); // waits only for what the sort did not cover
    GpuLcsSource& src = *held;
    double t1b = now_s();
    src.upload_ordered(s.codes.data(), s.offsets, in_of); // the records as they were read; the engine gathers the working order's on the device
    if (spare_ready.valid()) spare_ready.get();
    double t2 = now_s();
    if (t) t->rss_upload_kb = resident_kb();
    // The residues are on the device now; they go back to the system AFTER the tree stage, while the Newick is made.  Released
    // here, beside the levels of the heuristics, the unmapping of 765 MB (3 x 10^6 sequences) disturbed every thread of the tree
    // stage: 0.86-0.99 s against 0.71-0.89 s, and the whole command was 0.14 s slower (profiles/c5_stage_r06.txt;
    // FAMSA_HOST_TEST release_early / release_never: the other two ways).
    const int release_when = host_test("release_never") ? 2 : host_test("release_early") ? 0 : 1;
    if (consumable && release_when == 0) release_in_background(consumable->codes);
    struct rusage ru0, ru1;
    getrusage(RUSAGE_SELF, &ru0);
    std::string nwk = guide_tree_newick(s, w, src, opt, t, spare.empty() ? nullptr : &spare);
    if (consumable && release_when == 1) release_in_background(consumable->codes);
    if (profile_on()) { // how much of the tree stage the host's cores were busy
        getrusage(RUSAGE_SELF, &ru1);
        auto sec = [](const timeval& a, const timeval& b) { return (double)(b.tv_sec - a.tv_sec) + 1e-6 * (double)(b.tv_usec - a.tv_usec); };
        fprintf(stderr, "tree stage: %.3f s wall, host CPU %.3f s user + %.3f s system, %ld voluntary / %ld involuntary context switches\n",
                now_s() - t2, sec(ru0.ru_utime, ru1.ru_utime), sec(ru0.ru_stime, ru1.ru_stime), ru1.ru_nvcsw - ru0.ru_nvcsw,
                ru1.ru_nivcsw - ru0.ru_nivcsw);
    }
    if (t) {
        t->sort_s = t1 - t0;
        t->init_s = t1b - t1;
        t->upload_s = t2 - t1b;
        t->kernel_ms = src.kernel_ms_total();
        t->transport = src.transport();
    }
    if (g_abandon_engine_at_return && !profile_on()) (void)held.release();
    return nwk;
}
} // namespace

void dist_export_gpu(const SeqSet& s, int device, Distance dist, bool square, bool pid, const std::string& path,
                     Timings* t, EngineFuture* engine)
{
    EngineFuture own;
    if (!engine) {
        own = start_engine(device);
        engine = &own;
    }
    double t1 = now_s();
    std::unique_ptr<GpuLcsSource> held = engine->get();
    GpuLcsSource& src = *held;
    src.upload(s.codes.data(), s.offsets); // input order, no sort, no dedup: the set as it was read
    double t2 = now_s();
    src.keep_text_buffers = g_abandon_engine_at_return && !profile_on();
    write_distance_csv(src, s.ids, dist, square, pid, path);
    double t3 = now_s();
    if (t) {
        t->upload_s = t2 - t1;
        t->tree_s = t3 - t2;
        t->kernel_ms = src.kernel_ms_total();
        t->transport = src.transport();
    }
    if (g_abandon_engine_at_return && !profile_on()) (void)held.release();
}

} // namespace famsa_host
