// lcsgpu_internal.h -- shared between the translation units of liblcsgpu.so (not installed):
// the context, its lanes, error reporting, launch planning entry points.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <future>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lcsgpu.h"
#include "lcs_kernels.h"

namespace lcsgpu_impl {

// thread-local error text of lcsgpu_last_error(); returns `code`
int fail(int code, const char* fmt, ...);
// numeric tuning knobs, LCSGPU_TUNE="key=value,key=value" (no alternate code paths behind them): clarans_slice_us (a launch of
// search chains ends after that long and the chains continue in the next: 0 = none, a test aid), clarans_draws (step positions
// drawn for a sample shape before its first launch, 65536), assign_batch_kb (LCS rectangles of one launch of the batched seed
// assignment, 2 GB), upgma_spare (spare slots of the UPGMA matrix, n / 10)
int tune_int(const char* key, int dflt);

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(LCSGPU_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),    \
                        __FILE__, __LINE__);                                                    \
    } while (0)

// grow-only device / pinned-host buffers
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool owned = true; // false: a slice of somebody else's allocation (adopt); outgrown, it is replaced by an allocation of its own
    // hipFree waits for the whole device.  A buffer that grows while other host threads keep the GPU fed (the FRONT lane of
    // a level-by-level caller beside the leaf batches: 166 ms for one such free at 3 x 10^6 sequences) keeps what it has
    // outgrown until release() instead -- at most as much again as it ends up with.
    bool keep_outgrown = false;
    std::vector<void*> outgrown;
    void adopt(void* slice, size_t n)
    {
        release();
        p = slice;
        cap = n;
        owned = false;
    }
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p && owned) {
            if (keep_outgrown) outgrown.push_back(p);
            else (void)hipFree(p);
        }
        p = nullptr;
        cap = 0;
        owned = true;
        // 25% slack so that slowly growing requests do not reallocate every time -- but not on the
        // multi-GB result triangles, where the slack alone could be what does not fit
        size_t want = n + (n < ((size_t)256 << 20) ? n / 4 : 0) + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release()
    {
        if (p && owned) (void)hipFree(p);
        for (void* q : outgrown) (void)hipFree(q);
        outgrown.clear();
        p = nullptr;
        cap = 0;
        owned = true;
    }
};
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool keep_outgrown = false; // as DevBuf's
    std::vector<void*> outgrown;
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) {
            if (keep_outgrown) outgrown.push_back(p);
            else (void)hipHostFree(p);
        }
        p = nullptr;
        cap = 0;
        size_t want = n + n / 4 + 256;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release()
    {
        if (p) (void)hipHostFree(p);
        for (void* q : outgrown) (void)hipHostFree(q);
        outgrown.clear();
        p = nullptr;
        cap = 0;
    }
};

} // namespace lcsgpu_impl

// A lane = one HIP stream with its own staging / result buffers.  Host-memory calls (rect,
// triangle, triangle over ids) take any free lane, so several host threads -- the reference runs
// one CLCSBP per worker thread -- get their small LCS requests executed concurrently instead of
// queueing behind one stream; device-memory calls and the tree reducers always use lane 0, whose
// stream is the one lcsgpu_stream() hands out.
constexpr int MAX_LANES = 64;
// Beyond the ordinary lanes: slot MAX_LANES = the FRONT lane (LaneGuard::FRONT).
constexpr int LANE_SLOTS = MAX_LANES + 1;
struct Lane {
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    hipEvent_t ev_done = nullptr; // blocking-sync event: host-memory calls sleep on it instead of spinning
    hipStream_t copy_stream = nullptr; // large host-buffer results leave in slices while the next slice is computed
    lcsgpu_impl::DevBuf d_plan, d_out, d_carry;
    lcsgpu_impl::DevBuf d_work, d_draws; // work areas of the batched calls (CLARANS state; seeds of an assignment)
    lcsgpu_impl::PinBuf h_plan, h_small;
    bool plan_in_flight = false;
    int last_launches = 0;
    bool timing_valid = false;
    bool busy = false;
    // Lanes are created when first needed: a stream pair, its events and (with one hardware queue per lane)
    // its queue cost ~20 ms each -- 0.25 s for all 16 at start-up, most of what a small input paid in total.
    bool created = false;
    bool unusable = false; // creation failed: never offered again
};


struct TextExport; // lcsgpu_text.hip: the state of lcsgpu_dist_text_begin .. _end

// Who may have kernels on the chip: a wave of CLARANS chains (lcsgpu_clarans_batch: one workgroup per sample, each a long
// chain of short dependent phases) wants the CUs to itself -- bulk LCS launches of other host threads on the same CUs make
// every chain, and so the wave, several times slower (3 x 10^6 sequences: 278 chains 0.39 s beside the leaves' LCS batches,
// 0.04 s alone).  Bulk LCS calls hold the gate shared while their kernels run (not while their results travel), a wave holds
// it exclusively; a waiting wave goes first.  (Keeping 64 CUs for small waves by CU-masked streams instead -- the bulk
// launches masked to the other 192 -- did not keep the chains at their pace: 46-60 ms against 24, profiles/c5_levels_r06.txt.)
struct ComputeGate {
    std::mutex mu;
    std::condition_variable cv;
    int shared = 0, waiting_exclusive = 0;
    bool exclusive = false;
    void lock_shared()
    {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !exclusive && waiting_exclusive == 0; });
        ++shared;
    }
    void unlock_shared()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            --shared;
        }
        cv.notify_all();
    }
    void lock()
    {
        std::unique_lock<std::mutex> lk(mu);
        ++waiting_exclusive;
        cv.wait(lk, [&] { return !exclusive && shared == 0; });
        --waiting_exclusive;
        exclusive = true;
    }
    void unlock()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            exclusive = false;
        }
        cv.notify_all();
    }
};

struct lcsgpu_ctx {
    int device = 0;
    TextExport* text = nullptr;
    ComputeGate gate;
    // lcsgpu_clarans_batch: the samples' work areas (17 MB for 2000 members) come out of chunks that are kept from call to
    // call and only ever added to -- a fresh multi-GB hipMalloc is ~26 ms per GB, and a free waits for the whole device
    std::vector<lcsgpu_impl::DevBuf> sample_chunks;
    std::mutex mu; // guards the lane table
    std::condition_variable cv;
    std::vector<Lane> lanes;  // MAX_LANES slots; a slot costs nothing until its lane is created
    int lane_limit = 16;      // lanes [0, lane_limit) are handed out (lcsgpu_reserve_lanes raises it)

    // uploaded set (read-only while any lane is busy)
    int32_t n = -1;
    uint32_t max_len = 0;
    std::vector<uint32_t> lens;
    std::vector<uint8_t> quirk; // ref needs the literal (V2 < V) carry rule
    // per sequence: the half-word class its kernels are instantiated for (lcsgpu::h_class, 0 = beyond 2048 residues) | 0x80 if
    // quirk -- what the planners of the batched calls ask per id: one byte out of a table that stays in the host's cache
    // (lens + quirk: two cache misses per id, 3 x 10^6 sequences)
    std::vector<uint8_t> ref_class;
    lcsgpu_impl::DevBuf d_tiles, d_tile_base, d_lens, d_pow, d_powf, d_masks, d_mask_base;
    lcsgpu_impl::DevBuf d_minlen; // shortest sequence per aligned block of 16 vertices [ceil(n/16)], then of 1024 [ceil(n/1024)]
    const uint32_t* minlen16() const { return (const uint32_t*)d_minlen.p; }
    const uint32_t* minlen1024() const { return (const uint32_t*)d_minlen.p + ((size_t)(n > 0 ? n : 0) + 15) / 16; }

    // scratch of the lane-0 tree reducers
    lcsgpu_impl::DevBuf d_prim, d_qrows, d_qcols, d_dist;
    // sharded MST (lcsgpu_mst_shard_*): the replicated component state and this context's row block,
    // alive from _begin to the next _begin / upload
    lcsgpu_impl::DevBuf d_mst;
    lcsgpu_impl::DevBuf d_gather; // lcsgpu_multi_mst_prim: 2 x n_ctx x n keys, the slots the contexts push their keys into
    struct MstShard {
        bool active = false;
        lcsgpu::BoruvkaArgs b{};
        int elem = 2;
        int32_t found = 0; // MST edges recorded so far
        int rounds = 0;
        // how the local half of a round is done: by passes over the block's resident triangle (b.tri), or -- b.tri ==
        // NULL -- by recomputing the block's LCS values with the fold fused into the launch (O(n) memory);
        // fused_ready: the launch that filled b.tri has already done round 0's local half
        bool fused_ready = false;
        // passes_stand: row_best / part hold the passes' results of the round before, and the labels have only merged since
        // (a device merge; labels handed in by the caller end it): records whose edges still cross stand (BoruvkaArgs::keep)
        bool passes_stand = false;
    } mst;
    double total_kernel_ms = 0; // completed host-memory calls
};

namespace lcsgpu_impl {

struct LastCall { // timing of this thread's most recent call, for lcsgpu_last_kernel_ms
    lcsgpu_ctx* ctx = nullptr;
    std::vector<lcsgpu_ctx*> also; // the other contexts of a multi-context call: each answers for its own lane 0
    bool pending_on_lane0 = false;
    double ms = 0;
    int launches = 0;
};
extern thread_local LastCall g_last;

// streams / events of one lane (lcsgpu_api.hip); false on failure
bool create_lane(lcsgpu_ctx* ctx, Lane& l, bool high_priority = false);

// RAII ownership of one lane (index 0 on request, else any free one) or of all lanes.
class LaneGuard {
public:
    // FRONT: the context's one lane on a HIGH-PRIORITY stream (slot MAX_LANES, created on first use): the calls a
    // level-by-level caller waits for one after the other (lcsgpu_clarans_batch, lcsgpu_assign_seeds_batch) -- their
    // workgroups go before the pending ones of the bulk requests other host threads have queued on the ordinary lanes
    enum Which { ANY, LANE0, ALL, FRONT };
    LaneGuard(lcsgpu_ctx* ctx, Which which) : ctx_(ctx), which_(which)
    {
        std::unique_lock<std::mutex> lk(ctx->mu);
        if (which == ALL) {
            ctx->cv.wait(lk, [&] {
                for (auto& l : ctx->lanes) if (l.busy) return false;
                return true;
            });
            for (auto& l : ctx->lanes) l.busy = true;
            idx_ = 0;
        } else if (which == LANE0) {
            ctx->cv.wait(lk, [&] { return !ctx->lanes[0].busy; });
            ctx->lanes[0].busy = true;
            idx_ = 0;
        } else if (which == FRONT) {
            Lane& l = ctx->lanes[MAX_LANES];
            ctx->cv.wait(lk, [&] { return !l.busy; });
            l.busy = true;
            idx_ = MAX_LANES;
            if (!l.created && !l.unusable) {
                lk.unlock();
                const bool ok = create_lane(ctx, l, true);
                lk.lock();
                l.created = ok;
                l.unusable = !ok;
            }
            if (l.unusable) { // no such stream on this runtime: an ordinary lane serves
                l.busy = false;
                ctx->cv.notify_all();
                which_ = ANY;
                acquire_any(lk);
            }
        } else {
            acquire_any(lk);
        }
    }
    ~LaneGuard()
    {
        {
            std::lock_guard<std::mutex> lk(ctx_->mu);
            if (which_ == ALL) for (auto& l : ctx_->lanes) l.busy = false;
            else ctx_->lanes[idx_].busy = false;
        }
        ctx_->cv.notify_all();
    }
    Lane& lane() { return ctx_->lanes[idx_]; }

private:
    void acquire_any(std::unique_lock<std::mutex>& lk)
    {
        lcsgpu_ctx* ctx = ctx_;
        {
            // a created free lane other than lane 0 (device-memory calls and the tree reducers queue there);
            // else a lane that does not exist yet -- unless lane 0 is free and nothing else has been needed
            // so far (a single-threaded caller never pays for a second lane); else lane 0; else wait
            for (;;) {
                int pick = -1, fresh = -1;
                bool others = false;
                for (size_t i = 1; i < (size_t)ctx->lane_limit; ++i) {
                    const Lane& l = ctx->lanes[i];
                    if (l.unusable) continue;
                    if (l.created) others = true;
                    if (l.created && !l.busy && pick < 0) pick = (int)i;
                    if (!l.created && !l.busy && fresh < 0) fresh = (int)i;
                }
                if (pick < 0 && !ctx->lanes[0].busy && (!others || fresh < 0)) pick = 0;
                if (pick < 0 && fresh >= 0 && (others || ctx->lanes[0].busy)) {
                    Lane& l = ctx->lanes[fresh];
                    l.busy = true; // reserved while it is being created, outside the lock
                    lk.unlock();
                    const bool ok = create_lane(ctx, l);
                    lk.lock();
                    if (ok) {
                        l.created = true;
                        idx_ = fresh;
                        break;
                    }
                    l.unusable = true;
                    l.busy = false;
                    ctx->cv.notify_all(); // a LaneGuard(ALL) waiter counts busy lanes: this one just stopped being busy
                    continue;
                }
                if (pick >= 0) {
                    ctx->lanes[pick].busy = true;
                    idx_ = pick;
                    break;
                }
                ctx->cv.wait(lk);
            }
        }
    }
    lcsgpu_ctx* ctx_;
    Which which_;
    int idx_ = 0;
};

// Reserve a large device buffer, reporting LCSGPU_E_NOMEM (not a HIP error) when it cannot fit.
int reserve_big(lcsgpu_ctx* ctx, DevBuf& buf, size_t bytes, const char* what);
int device_free_bytes(lcsgpu_ctx* ctx, size_t* out); // as reserve_big sees it (LCSGPU_FAKE_HBM_GB included)
// The n-1 tree edges in the order Prim's algorithm adds them from vertex 0 (host; in place).
int order_edges_like_prim(lcsgpu_mst_edge* edges, int32_t n);

// frees what lcsgpu_dist_text_begin set up (lcsgpu_text.hip); the caller holds lane 0 or all lanes
void text_release(lcsgpu_ctx* ctx);

// Core: plan + launch.  d_out is a device pointer.
// fuse != NULL (triangle mode, contiguous rows and columns): the launches also fold their results into the
// per-vertex best-edge records (lcs_kernels.h, FuseArgs); d_out may then be NULL (nothing is stored).
int run_rows(lcsgpu_ctx* ctx, Lane& L, int mode, const int32_t* ref_ids, int32_t ref_begin, int32_t n_refs,
             const int32_t* col_ids, int32_t col_begin, int32_t n_cols, void* d_out, int64_t ld,
             int64_t out_offset, int elem_size, int64_t first_row = 0, const lcsgpu::FuseArgs* fuse = nullptr);
// which instantiation the refs of half-word class h run in (target[h] >= h), given wgs[h] = the workgroups class h would
// have on its own, h = 1 .. 64: small neighbouring classes share a launch (lcsgpu_api.hip)
void merge_small_classes(const double* wgs, int* target);
// After a host-memory call has been synchronised: account its kernel time.
void finish_host_call(lcsgpu_ctx* ctx, Lane& L);
// a host-memory call that keeps its own events (lcsgpu_text.hip): this thread's answer to lcsgpu_last_kernel_ms
void note_host_call(lcsgpu_ctx* ctx, double ms, int launches);
// A *_dev call was queued on lane 0: its timing is read on demand.
// also_this: a further context of the same multi-context call (the first one is noted without the flag).
void note_async_call(lcsgpu_ctx* ctx, bool also_this = false);

} // namespace lcsgpu_impl
