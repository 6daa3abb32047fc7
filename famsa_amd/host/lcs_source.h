// lcs_source.h -- where the tree builders get oriented LCS lengths from.
// The product implementation (GpuLcsSource) is the C-ABI of include/lcsgpu.h; MatrixLcsSource
// serves a caller-supplied matrix so the host logic can be exercised without a GPU (tests feed
// it the oracle's matrix).  There is no CPU LCS implementation on this side of the boundary.
#pragma once
#include <atomic>
#include <cstdint>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "noinit.h"

struct lcsgpu_ctx;

namespace famsa_host {

// Test hooks of the host layer, FAMSA_HOST_TEST="name,name,key=value": prim_streaming (MST-Prim's O(n)-memory host form on
// any input), slink_from_mst (the MST -> SLINK conversion with Prim on the host), upgma_triangle (the leaf UPGMA's triangle
// walk instead of the square matrix), no_device_mst (as if the triangle did not fit the device), clarans_host (the CLARANS
// search on the host), threads=N (worker threads of the C test entry points), pool=N (threads of the FastTree recursion's task
// pool instead of twice the cores: measurements), csv_input_order (-dist_export asks for its rectangles with the columns as
// they were read instead of by length: measurements), leafmax=N (threads of that pool that may work on leaves while splits wait:
// measurements), release_early / release_never / no_spare_tree / no_level_scratch (host memory of the tree heuristics as it was
// before the second session of round 6: measurements).  Not read on any product default path.
bool host_test(const char* name);
int host_test_int(const char* key, int dflt);
// LCSGPU_PROFILE: stage / call statistics on stderr (the library prints its own under the same switch)
bool profile_on();

// LCS values as uint16 (all sequences <= 65535 residues) or uint32.
struct LcsBuf { // (a resize leaves new elements unset: whoever resizes fills all of them)
    bool wide = false;
    std::vector<uint16_t, NoInit<uint16_t>> v16;
    std::vector<uint32_t, NoInit<uint32_t>> v32;
    void resize(size_t n, bool wide_)
    {
        wide = wide_;
        v16.clear(); // (first: growing would otherwise move the old values to the new block)
        v32.clear();
        if (wide) v32.resize(n); else v16.resize(n);
    }
    size_t size() const { return wide ? v32.size() : v16.size(); }
    uint32_t operator[](size_t i) const { return wide ? v32[i] : v16[i]; }
    void* data() { return wide ? (void*)v32.data() : (void*)v16.data(); }
    int elem_size() const { return wide ? 4 : 2; }
};

class LcsSource {
public:
    virtual ~LcsSource() {}
    virtual int n() const = 0;
    virtual uint32_t length(int i) const = 0;
    // true if some sequence, used as the ref, can give an orientation-dependent value
    // (reference carry rule, SURVEY note Q): then LCS(ref=a,partner=b) != LCS(ref=b,partner=a) may hold
    virtual bool orientation_sensitive() const = 0;
    // rows [r0, r1) of the lower triangle, ref = row i, partner = column j < i; out[i*(i-1)/2 + j - r0*(r0-1)/2]
    virtual void triangle(int r0, int r1, LcsBuf& out) = 0;
    // out[r*n_cols + c] = LCS(ref = refs[r], partner = cols ? cols[c] : c)
    virtual void rect(const int* refs, int n_refs, const int* cols, int n_cols, LcsBuf& out) = 0;
    // lower triangle over an id list: out[k*(k-1)/2 + c] = LCS(ref = ids[k], partner = ids[c]), c < k
    virtual void triangle_ids(const int* ids, int n_ids, LcsBuf& out);
    // values need 32 bits (some sequence longer than 65535 residues); sources cache this
    virtual bool wide() const;
    // The whole lower triangle where the source already holds it in host memory (uint16, or uint32 if wide()):
    // element i*(i-1)/2 + j, j < i.  NULL = not resident; call triangle().
    virtual const void* triangle_view() const { return nullptr; }
    // a hint: requests will come from this many host threads at once (the FastTree pool)
    virtual void expect_threads(int /*n_threads*/) {}
    // Prim's MST computed by the source itself (the GPU engine does it on the device): n-1 edges
    // (from < to, distance) in the order they are added from vertex 0.  Returns false if the
    // source cannot (then the caller runs Prim on the host over triangle()/rect()).
    struct MstEdge { int32_t from, to; double dist; };
    // triangle_orientation: every distance from LCS(ref = larger id, partner = smaller id) (what SLINK
    // sees) instead of MSTPrim's LCS(ref = node just added, partner = candidate)
    virtual bool prim_edges(int /*distance_kind*/, std::vector<MstEdge>& /*edges*/, bool /*triangle_orientation*/) { return false; }
    // UPGMA computed by the source itself (device): children of internal nodes n..2n-2.
    virtual bool upgma_nodes(int /*distance_kind*/, bool /*modified*/, std::vector<int32_t>& /*left*/,
                             std::vector<int32_t>& /*right*/) { return false; }
    virtual bool nj_nodes(int /*distance_kind*/, std::vector<int32_t>& /*left*/, std::vector<int32_t>& /*right*/) { return false; }
    // The packed triangles of several id lists in one request: list g = ids[offsets[g] .. offsets[g+1]),
    // its triangle at out[sum_{h<g} m_h(m_h-1)/2 ...].  False = not offered; ask list by list.
    virtual bool triangles_batch(const int* /*ids*/, const int64_t* /*offsets*/, int /*n_groups*/, LcsBuf& /*out*/) { return false; }
    // Seed assignment of one FastTree evaluation done by the source itself: for r in order, a column moves to
    // seed first_k + r on a strictly smaller Transform<float> distance; dist / assign are updated in place.
    virtual bool assign_seeds(const int* /*seeds*/, int /*n_seeds*/, const int* /*cols*/, int /*n_cols*/, int /*distance_kind*/,
                              int /*first_k*/, float* /*dist*/, int* /*assign*/) { return false; }
    // The same for several samples at once (all splits of a level of the FastTree recursion): sample g =
    // ids[offsets[g] .. offsets[g + 1]) with n_medoids[g] medoids; medoids_out holds the samples' medoids one after the other.
    // False = not offered: nothing has been computed, ask sample by sample.
    virtual bool clarans_batch(const int* /*ids*/, const int64_t* /*offsets*/, int /*n_jobs*/, int /*distance_kind*/, const int* /*n_medoids*/,
                               int /*n_fixed*/, float /*explore_fraction*/, int /*num_local*/, int* /*medoids_out*/) { return false; }
    // Seed assignment of several evaluations at once, each from scratch: job g has the seeds seeds[seed_off[g] .. seed_off[g + 1])
    // and the columns cols[col_off[g] .. col_off[g + 1]); per column (dist / assign laid out like cols): the smallest
    // Transform<float> distance to a seed of its job and the number (0-based within the job) of the FIRST seed that attains it.
    virtual bool assign_seeds_batch(const int* /*seeds*/, const int64_t* /*seed_off*/, const int* /*cols*/, const int64_t* /*col_off*/, int /*n_jobs*/,
                                    int /*distance_kind*/, float* /*dist*/, int* /*assign*/) { return false; }
    // -dist_export rows as TEXT made by the source itself (the device formats them): text_begin returns the number of
    // independent units blocks can be in flight on (0 = not offered: the caller formats LCS values on the host); a unit takes
    // one block of rows [r0, r1) at a time: text_submit queues it, text_wait returns its bytes -- the rows one after the
    // other exactly as the file holds them -- valid until the unit's next submit.  ids as read (with their '>').
    virtual int text_begin(const std::vector<std::string>& /*ids*/, int /*distance_kind*/, bool /*square*/, bool /*pid*/) { return 0; }
    virtual void text_submit(int /*unit*/, int /*r0*/, int /*r1*/) {}
    virtual void text_wait(int /*unit*/, const char*& /*text*/, uint64_t& /*bytes*/) {}
    virtual void text_end() {}
    // CLARANS k-medoids over the sample `ids` computed by the source itself (device): medoids[k] =
    // member numbers 0..n_ids-1.  False = not offered for this shape; the caller runs the host search.
    virtual bool clarans(const int* /*ids*/, int /*n_ids*/, int /*distance_kind*/, int /*n_medoids*/, int /*n_fixed*/,
                         float /*explore_fraction*/, int /*num_local*/, int* /*medoids*/) { return false; }
};

// The MI355X engine.  Throws std::runtime_error if the library reports an error (no fallback).
// With several devices (one context each, all holding the uploaded set) the whole-set requests are tiled by
// row blocks over the GPUs (lcsgpu_multi_*), the batched requests of the FastTree recursion are split between
// them (leaf matrices by pair count, seed assignment by columns) and the small per-thread requests go round robin.
class GpuLcsSource : public LcsSource {
public:
    explicit GpuLcsSource(int device);
    explicit GpuLcsSource(const std::vector<int>& devices);
    ~GpuLcsSource() override;
    int n_devices() const { return (int)ctxs_.size(); }
    // the FastTree recursion calls from `n_threads` host threads: have the engine's lanes ready (lcsgpu_reserve_lanes)
    void expect_threads(int n_threads) override;
    void upload(const uint8_t* codes, const std::vector<uint64_t>& offsets); // codes[offsets[i] .. offsets[i + 1]) = sequence i
    // sequence k of the set = record order[k] of (codes, offsets) (lcsgpu_upload_ordered: no packed host copy)
    void upload_ordered(const uint8_t* codes, const std::vector<uint64_t>& offsets, const std::vector<int>& order);
    int n() const override { return (int)lens_.size(); }
    uint32_t length(int i) const override { return lens_[i]; }
    bool orientation_sensitive() const override { return sensitive_; }
    bool wide() const override { return wide_; }
    void triangle(int r0, int r1, LcsBuf& out) override;
    void rect(const int* refs, int n_refs, const int* cols, int n_cols, LcsBuf& out) override;
    void triangle_ids(const int* ids, int n_ids, LcsBuf& out) override;
    bool prim_edges(int distance_kind, std::vector<MstEdge>& edges, bool triangle_orientation) override;
    bool upgma_nodes(int distance_kind, bool modified, std::vector<int32_t>& left, std::vector<int32_t>& right) override;
    bool nj_nodes(int distance_kind, std::vector<int32_t>& left, std::vector<int32_t>& right) override;
    bool clarans(const int* ids, int n_ids, int distance_kind, int n_medoids, int n_fixed, float explore_fraction,
                 int num_local, int* medoids) override;
    bool triangles_batch(const int* ids, const int64_t* offsets, int n_groups, LcsBuf& out) override;
    bool assign_seeds(const int* seeds, int n_seeds, const int* cols, int n_cols, int distance_kind, int first_k, float* dist,
                      int* assign) override;
    bool clarans_batch(const int* ids, const int64_t* offsets, int n_jobs, int distance_kind, const int* n_medoids, int n_fixed,
                       float explore_fraction, int num_local, int* medoids_out) override;
    bool assign_seeds_batch(const int* seeds, const int64_t* seed_off, const int* cols, const int64_t* col_off, int n_jobs,
                            int distance_kind, float* dist, int* assign) override;
    int text_begin(const std::vector<std::string>& ids, int distance_kind, bool square, bool pid) override;
    void text_submit(int unit, int r0, int r1) override;
    void text_wait(int unit, const char*& text, uint64_t& bytes) override;
    void text_end() override;
    double kernel_ms_total() const { return kernel_ms_; }
    // several devices: lcsgpu_multi_transport's text (empty with one context)
    std::string transport() const;
    void add_kernel_ms(lcsgpu_ctx* ctx);

private:
    void upload_records(const uint8_t* codes, const std::vector<uint64_t>& offsets, const int* order, int32_t n);
    void check(int rc, const char* what);
    lcsgpu_ctx* pick(); // the context for a small request: round robin over the devices
    std::vector<lcsgpu_ctx*> ctxs_;
    lcsgpu_ctx* ctx_ = nullptr; // = ctxs_[0]
    std::atomic<unsigned> next_{0};
    std::vector<uint32_t> lens_;
    bool sensitive_ = false, wide_ = false;
    double kernel_ms_ = 0;
    int text_slots_ = 0; // per context, between text_begin and text_end
public:
    // a caller that is about to abandon the engine (a command-line tool at its end) lets text_end leave the slots' buffers
    // to the engine's destruction: freeing pinned and device memory is ~15 ms it would only wait for
    bool keep_text_buffers = false;
private:
    std::mutex mu_; // the tree builders may call from several threads
    // call statistics (printed at destruction when FAMSA_GPU_PROFILE is set)
    struct CallStat { long calls = 0; double seconds = 0; double pairs = 0; } st_rect_, st_tri_, st_triids_, st_clarans_, st_batch_, st_assign_;
    // contiguous job ranges of about equal weight, one per context; fn(context index, first job, end job) on a thread each
    void over_contexts(int n_jobs, const std::vector<double>& weight, const std::function<void(int, int, int)>& fn);
    void note(CallStat& s, double sec, double pairs);
};

// A full oriented square matrix supplied by the caller: m[ref*n + partner].
class MatrixLcsSource : public LcsSource {
public:
    MatrixLcsSource(int n, const uint32_t* lens, const uint32_t* square);
    int n() const override { return n_; }
    uint32_t length(int i) const override { return lens_[i]; }
    bool orientation_sensitive() const override { return sensitive_; }
    bool wide() const override { return wide_; }
    void triangle(int r0, int r1, LcsBuf& out) override;
    void rect(const int* refs, int n_refs, const int* cols, int n_cols, LcsBuf& out) override;
    // served from the matrix, so the batched-leaf path of the FastTree recursion runs without a GPU too
    bool triangles_batch(const int* ids, const int64_t* offsets, int n_groups, LcsBuf& out) override;

private:
    int n_;
    std::vector<uint32_t> lens_, m_;
    bool sensitive_, wide_ = false;
};

} // namespace famsa_host
