// gpu_lcsbp.cpp -- INTEGRATION.md section 2b, COMPILED: the reference's SIMD dispatcher replaced by the GPU engine.
//
// TEST INFRASTRUCTURE (lives under oracle/, built into oracle/_ref/libfamsa_gpuref.so, loaded only by tests/): it
// proves that the C-ABI of include/lcsgpu.h is a drop-in at the seam SURVEY.md section 2 row 5 names -- "the seam
// where a GPU back-end replaces CPU lanes": class CLCSBP (reference src/lcs/lcsbp.h:16-51).  This file IMPLEMENTS
// that class -- the constructor and the four GetLCSBP overloads the reference's header declares
// (lcs/lcsbp.h:36-46) -- on top of liblcsgpu.so, and oracle/Makefile links it with the reference's own guide-tree
// objects (tree/MSTPrim.cpp, SingleLinkage.cpp, UPGMA.cpp, NeighborJoining.cpp, DistanceCalculator.cpp,
// FastTree.cpp, Clustering.cpp, compiled from /root/reference) IN PLACE OF the reference's lcs/lcsbp.cpp,
// lcs/lcsbp_classic.cpp and simd/lcsbp_avx*_intr.cpp.  The library that results contains no CPU LCS code at all:
// every LCS length any of the six generators sees -- MSTPrim's per-step batches (tree/MSTPrim.cpp:478-485),
// SLINK's and UPGMA's rows (SingleLinkage.cpp:58-81, UPGMA.cpp:75-109), the CSV rows of -dist_export
// (DistanceCalculator.cpp:28-82), FastTree's rectangles, sample matrices and leaf matrices
// (FastTree.cpp:309-324, 347, 385, 415) -- comes out of a HIP kernel, while the batch templates
// (AbstractTreeGenerator.hpp), the Transform functors, the tree algorithms, CLARANS and the writers stay the
// reference's object code.
//
// How the seam is served.  GetLCSBP(ref, p1..p8, out) asks for 4 or 8 LCS lengths of one ref; a GPU wants >= 10^4
// pairs per call.  The reference itself caches per-ref state behind this interface (the AVX2 variant keeps the pair-
// mask table of the last ref, keyed by sequence_no: simd/lcsbp_avx2_intr.cpp:42); this implementation caches the
// ref's whole ROW instead: the first request for ref r computes LCS(ref = r, partner = j) for every sequence j of
// the bound set in one lcsgpu_lcs_rect call (orientation kept: r is the bit-mask side), later requests are look-ups.
// Sequences are identified by CSequence::sequence_no (msa.cpp:559-561: rank in the working set) and, for the
// CSequenceView partners of MSTPrim (MSTPrim.cpp:837-852), by their data pointer.
//
// What a maintainer adds for this to be live in FAMSA: instruction_set_t::gpu selecting this implementation of
// CLCSBP, and one call of gpu_lcsbp_bind(sequences) in AbstractTreeGenerator::operator() (AbstractTreeGenerator.cpp:
// 25-32, where the tree is sized before run()) -- here ref_harness.cpp's pre-run hook makes that call.
#include "core/sequence.h"
#include "lcs/lcsbp.h"

#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <list>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <unordered_map>
#include <vector>

#include "../include/lcsgpu.h"

extern "C" void ref_set_pre_run_hook(void (*hook)(CSequence* const* seqs, int n, void* user), void* user);

namespace {

struct Row {
    std::vector<uint32_t> lcs; // LCS(ref = this row's sequence, partner = j), j < n
    bool ready = false;
    bool failed = false;
};

struct Binding {
    std::mutex mu;
    std::condition_variable cv;
    lcsgpu_ctx* ctx = nullptr;
    int n = 0;
    uint64_t epoch = 0; // bumped by every bind: per-thread caches of an earlier set are stale
    std::unordered_map<const void*, int> by_data; // CSequence::data -> id (the CSequenceView partners of MSTPrim)
    std::vector<uint32_t> lens;
    // rows in use, shared by the worker threads (MSTPrim: all workers ask for the same ref; SLINK / UPGMA: one row each)
    std::unordered_map<int, std::shared_ptr<Row>> rows;
    std::list<int> order; // oldest first
    size_t cap = 64;
    std::atomic<uint64_t> n_rows{0}, n_values{0};
};
Binding g;

struct ThreadCache {
    uint64_t epoch = ~0ull;
    int ref = -1;
    std::shared_ptr<Row> row;
};
thread_local ThreadCache t_cache;

[[noreturn]] void die(const char* what)
{
    // the reference's interface is void and never fails; an engine error here is a broken test set-up
    fprintf(stderr, "gpu_lcsbp: %s: %s\n", what, lcsgpu_last_error());
    abort();
}

const std::vector<uint32_t>& row_of(const CSequence* ref)
{
    const int id = ref->sequence_no;
    if (t_cache.epoch == g.epoch && t_cache.ref == id) return t_cache.row->lcs;
    std::shared_ptr<Row> row;
    bool compute = false;
    {
        std::unique_lock<std::mutex> lk(g.mu);
        if (!g.ctx) die("GetLCSBP before gpu_lcsbp_bind");
        if (id < 0 || id >= g.n || g.lens[id] != ref->length) die("the ref is not a sequence of the bound set");
        auto it = g.rows.find(id);
        if (it != g.rows.end()) {
            row = it->second;
            g.cv.wait(lk, [&] { return row->ready || row->failed; });
        } else {
            row = std::make_shared<Row>();
            g.rows[id] = row;
            g.order.push_back(id);
            while (g.order.size() > g.cap) { // whoever still holds an evicted row keeps it alive through its shared_ptr
                g.rows.erase(g.order.front());
                g.order.pop_front();
            }
            compute = true;
        }
    }
    if (compute) {
        row->lcs.resize((size_t)g.n);
        const int32_t r = id;
        const int rc = lcsgpu_lcs_rect(g.ctx, &r, 0, 1, nullptr, 0, g.n, row->lcs.data(), g.n, 4);
        {
            std::lock_guard<std::mutex> lk(g.mu);
            row->ready = rc == LCSGPU_OK;
            row->failed = rc != LCSGPU_OK;
        }
        g.cv.notify_all();
        if (rc != LCSGPU_OK) die("lcsgpu_lcs_rect");
        g.n_rows.fetch_add(1, std::memory_order_relaxed);
    }
    if (row->failed) die("lcsgpu_lcs_rect (another thread's request)");
    t_cache.epoch = g.epoch;
    t_cache.ref = id;
    t_cache.row = row;
    return row->lcs;
}

inline int id_of(const CSequence* s) { return s->sequence_no; }
inline int id_of(const CSequenceView* v)
{
    auto it = g.by_data.find((const void*)v->data); // read-only after bind
    if (it == g.by_data.end()) die("a CSequenceView that does not point into the bound set");
    return it->second;
}

template <class P>
void serve(CSequence* ref, P* const* partners, int n_partners, uint32_t* dist)
{
    const std::vector<uint32_t>& row = row_of(ref);
    for (int k = 0; k < n_partners; ++k)
        if (partners[k]) { // a null partner = the padding of a partial group (lcs/lcsbp.cpp:166-176): its slot is not read
            const int j = id_of(partners[k]);
            if (j < 0 || j >= g.n) die("partner id out of range");
            dist[k] = row[(size_t)j];
        }
    g.n_values.fetch_add((uint64_t)n_partners, std::memory_order_relaxed);
}

void bind_hook(CSequence* const* seqs, int n, void* user);

struct Registrar {
    Registrar() { ref_set_pre_run_hook(bind_hook, nullptr); }
} registrar;

} // namespace

extern "C" {

// Upload the working set (unpadded symbol codes, ids = sequence_no = position) and start a fresh row cache.
int gpu_lcsbp_bind(CSequence* const* seqs, int n, int device)
{
    std::lock_guard<std::mutex> lk(g.mu);
    if (!g.ctx && lcsgpu_create(device, &g.ctx) != LCSGPU_OK) return -1;
    std::vector<uint64_t> offsets((size_t)n + 1, 0);
    for (int i = 0; i < n; ++i) {
        if (seqs[i]->sequence_no != i) return -2; // the reference numbers the working set 0..n-1 (msa.cpp:559-561)
        offsets[i + 1] = offsets[i] + seqs[i]->length;
    }
    std::vector<uint8_t> codes(offsets[n] ? offsets[n] : 1);
    g.by_data.clear();
    g.lens.assign((size_t)n, 0);
    for (int i = 0; i < n; ++i) {
        for (uint32_t p = 0; p < seqs[i]->length; ++p) codes[offsets[i] + p] = (uint8_t)seqs[i]->data[p];
        g.by_data[(const void*)seqs[i]->data] = i;
        g.lens[i] = seqs[i]->length;
    }
    if (lcsgpu_upload(g.ctx, codes.data(), offsets.data(), n) != LCSGPU_OK) return -3;
    g.n = n;
    g.rows.clear();
    g.order.clear();
    ++g.epoch;
    return 0;
}

// rows computed on the GPU / LCS values handed out since the library was loaded (the tests check that the
// generators really were served from here)
void gpu_lcsbp_stats(uint64_t* rows, uint64_t* values)
{
    *rows = g.n_rows.load();
    *values = g.n_values.load();
}

} // extern "C"

namespace {
void bind_hook(CSequence* const* seqs, int n, void*)
{
    const char* dev = getenv("GPU_LCSBP_DEVICE");
    const int rc = gpu_lcsbp_bind(seqs, n, dev ? atoi(dev) : 0);
    if (rc) {
        fprintf(stderr, "gpu_lcsbp_bind failed (%d): %s\n", rc, lcsgpu_last_error());
        throw std::runtime_error("gpu_lcsbp_bind failed");
    }
}
} // namespace

// ---- class CLCSBP as declared in the reference's lcs/lcsbp.h, implemented over the engine -----------------------
// No CPU kernel object is created: the shared_ptr members of the reference's declaration stay empty.
CLCSBP::CLCSBP(instruction_set_t _instruction_set) { instruction_set = _instruction_set; }

void CLCSBP::GetLCSBP(CSequence* seq0, CSequence* seq1, CSequence* seq2, CSequence* seq3, CSequence* seq4, uint32_t* dist)
{
    CSequence* p[4] = {seq1, seq2, seq3, seq4};
    serve(seq0, p, 4, dist);
}

void CLCSBP::GetLCSBP(CSequence* seq0, CSequenceView* sv1, CSequenceView* sv2, CSequenceView* sv3, CSequenceView* sv4,
                      uint32_t* dist)
{
    CSequenceView* p[4] = {sv1, sv2, sv3, sv4};
    serve(seq0, p, 4, dist);
}

void CLCSBP::GetLCSBP(CSequence* seq0, CSequence* seq1, CSequence* seq2, CSequence* seq3, CSequence* seq4, CSequence* seq5,
                      CSequence* seq6, CSequence* seq7, CSequence* seq8, uint32_t* dist)
{
    CSequence* p[8] = {seq1, seq2, seq3, seq4, seq5, seq6, seq7, seq8};
    serve(seq0, p, 8, dist);
}

void CLCSBP::GetLCSBP(CSequence* seq0, CSequenceView* sv1, CSequenceView* sv2, CSequenceView* sv3, CSequenceView* sv4,
                      CSequenceView* sv5, CSequenceView* sv6, CSequenceView* sv7, CSequenceView* sv8, uint32_t* dist)
{
    CSequenceView* p[8] = {sv1, sv2, sv3, sv4, sv5, sv6, sv7, sv8};
    serve(seq0, p, 8, dist);
}
