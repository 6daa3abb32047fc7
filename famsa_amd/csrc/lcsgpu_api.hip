// lcsgpu_api.hip -- the C-ABI of include/lcsgpu.h over the gfx950 kernels.
// Context, HBM layout of the uploaded sequence set, launch planning.  No CPU compute path:
// every LCS value this library returns was produced by a HIP kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cmath>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>
#include <random>
#include <string>
#include <vector>

#include "../../include/lcsgpu.h"
#include "lcs_kernels.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

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
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = n + n / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};
struct PinBuf {
    void* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
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
        p = nullptr;
        cap = 0;
    }
};

} // namespace

// A lane = one HIP stream with its own staging / result buffers.  Host-memory calls (rect,
// triangle, triangle over ids) take any free lane, so several host threads -- the reference runs
// one CLCSBP per worker thread -- get their small LCS requests executed concurrently instead of
// queueing behind one stream; device-memory calls and the tree reducers always use lane 0, whose
// stream is the one lcsgpu_stream() hands out.
struct Lane {
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    hipEvent_t ev_done = nullptr; // blocking-sync event: host-memory calls sleep on it instead of spinning
    DevBuf d_plan, d_out, d_carry;
    DevBuf d_work, d_draws; // CLARANS state and its pre-drawn step positions
    PinBuf h_plan, h_small;
    bool plan_in_flight = false;
    int last_launches = 0;
    bool timing_valid = false;
    bool busy = false;
};


// Local searches of several host threads advanced together (lcs_kernels.h, ClaransBatch): every
// search joins with its device state ready; whichever owner finds no driver becomes the driver and
// enqueues the rounds for ALL joined searches, looking at their done flags every `rounds_per_look`
// rounds; a driver whose own search has finished hands the role to one of the remaining owners.
struct ClaransJob {
    lcsgpu::ClaransArgs a;
    std::mt19937* gen_positions = nullptr; // the owner's position generator (Clustering.cpp:44)
    std::vector<int32_t>* draws = nullptr; // its output so far, as accepted draws
    DevBuf* d_draws = nullptr;
    int32_t p_host = 0;
    int32_t state[16] = {0};
    bool done = false;
    int rc = LCSGPU_OK;
    std::string error;
};
struct ClaransBatcher {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<ClaransJob*> joined;
    bool driver_present = false;
    hipStream_t stream = nullptr;
    hipEvent_t ev = nullptr;
    PinBuf h_states;
    // LCSGPU_PROFILE: looks and seconds by number of searches in the batch
    long prof_looks[lcsgpu::CLARANS_MAX_BATCH + 1] = {0};
    double prof_seconds[lcsgpu::CLARANS_MAX_BATCH + 1] = {0};
};

struct lcsgpu_ctx {
    int device = 0;
    std::mutex mu; // guards the lane table
    std::condition_variable cv;
    std::vector<Lane> lanes;

    // uploaded set (read-only while any lane is busy)
    int32_t n = -1;
    uint32_t max_len = 0;
    std::vector<uint32_t> lens;
    std::vector<uint8_t> quirk; // ref needs the literal (V2 < V) carry rule
    DevBuf d_tiles, d_tile_base, d_lens, d_pow, d_powf;

    // scratch of the lane-0 tree reducers
    DevBuf d_prim, d_qrows, d_qcols, d_dist;
    double total_kernel_ms = 0; // completed host-memory calls
    // searches are spread over a few independent batches (each its own stream and driver): rounds of
    // different batches overlap on the GPU, which hides part of a round's memory latency
    std::vector<ClaransBatcher> clarans_groups;
    std::atomic<unsigned> clarans_next{0};
};

namespace {

thread_local struct { // timing of this thread's most recent call, for lcsgpu_last_kernel_ms
    lcsgpu_ctx* ctx = nullptr;
    bool pending_on_lane0 = false;
    double ms = 0;
    int launches = 0;
} g_last;

// RAII ownership of one lane (index 0 on request, else any free one) or of all lanes.
class LaneGuard {
public:
    enum Which { ANY, LANE0, ALL };
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
        } else {
            ctx->cv.wait(lk, [&] {
                for (size_t i = ctx->lanes.size(); i-- > 0;) // prefer the higher lanes, keep lane 0 free
                    if (!ctx->lanes[i].busy) { idx_ = (int)i; return true; }
                return false;
            });
            ctx->lanes[idx_].busy = true;
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
    lcsgpu_ctx* ctx_;
    Which which_;
    int idx_ = 0;
};

} // namespace

namespace {

using lcsgpu::RowsArgs;

struct RefItem {
    int32_t id;
    int64_t row;
};

struct Bucket {
    int bv;
    bool quirk;
    std::vector<RefItem> items;
};

// Split the refs by instantiated kernel (word-count class, quirk flag), keeping order.
int make_buckets(lcsgpu_ctx* ctx, const int32_t* ref_ids, int32_t ref_begin, int32_t n_refs,
                 int64_t row0, std::vector<Bucket>& out)
{
    int index_of[160];
    std::fill(index_of, index_of + 160, -1);
    for (int32_t k = 0; k < n_refs; ++k) {
        const int32_t id = ref_ids ? ref_ids[k] : ref_begin + k;
        if (id < 0 || id >= ctx->n)
            return fail(LCSGPU_E_INVALID, "ref id %d out of range [0,%d)", id, ctx->n);
        const bool q = ctx->quirk[id] != 0;
        // bv = instantiated half-word count; 0 = the long-sequence kernel (> 2048 residues)
        const int bv = q ? lcsgpu::quirk_h_class(ctx->lens[id]) : lcsgpu::h_class(ctx->lens[id]);
        const int key = bv * 2 + (q ? 1 : 0);
        if (index_of[key] < 0) {
            index_of[key] = (int)out.size();
            out.push_back(Bucket{bv, q, {}});
        }
        out[index_of[key]].items.push_back(RefItem{id, row0 + k});
    }
    return LCSGPU_OK;
}

bool contiguous(const Bucket& b)
{
    for (size_t k = 1; k < b.items.size(); ++k)
        if (b.items[k].id != b.items[0].id + (int32_t)k || b.items[k].row != b.items[0].row + (int64_t)k)
            return false;
    return true;
}

// Core: plan + launch.  d_out is a device pointer.
int run_rows(lcsgpu_ctx* ctx, Lane& L, int mode, const int32_t* ref_ids, int32_t ref_begin, int32_t n_refs,
             const int32_t* col_ids, int32_t col_begin, int32_t n_cols, void* d_out, int64_t ld,
             int64_t out_offset, int elem_size, int64_t first_row = 0)
{
    if (ctx->n < 0) return fail(LCSGPU_E_STATE, "no sequence set uploaded");
    if (elem_size != 2 && elem_size != 4) return fail(LCSGPU_E_INVALID, "elem_size must be 2 or 4");
    if (elem_size == 2 && ctx->max_len > 65535)
        return fail(LCSGPU_E_INVALID, "uint16 output needs all sequences <= 65535 residues");
    if (n_refs < 0 || n_cols < 0) return fail(LCSGPU_E_INVALID, "negative count");
    L.last_launches = 0;
    L.timing_valid = false;
    if (n_refs == 0 || n_cols == 0) return LCSGPU_OK;
    if (!col_ids && (col_begin < 0 || (int64_t)col_begin + n_cols > ctx->n))
        return fail(LCSGPU_E_INVALID, "column range out of bounds");
    if (col_ids)
        for (int32_t c = 0; c < n_cols; ++c)
            if (col_ids[c] < 0 || col_ids[c] >= ctx->n)
                return fail(LCSGPU_E_INVALID, "col id %d out of range", col_ids[c]);

    std::vector<Bucket> buckets;
    int rc = make_buckets(ctx, ref_ids, ref_begin, n_refs, first_row, buckets);
    if (rc) return rc;

    // staging: [col_ids][per non-contiguous bucket: rows(int64) then ids(int32)]
    size_t bytes = 0;
    auto align8 = [](size_t x) { return (x + 7) & ~(size_t)7; };
    const size_t col_off = 0;
    if (col_ids) bytes += align8((size_t)n_cols * 4);
    std::vector<size_t> row_off(buckets.size(), 0), id_off(buckets.size(), 0), pre_off(buckets.size(), 0);
    std::vector<char> is_contig(buckets.size(), 0);
    std::vector<std::vector<int32_t>> tri_prefix(buckets.size());
    std::vector<int> refs_per_wg(buckets.size(), 0);
    for (size_t b = 0; b < buckets.size(); ++b) {
        // triangle: about half of the column blocks of a ref tile lie below the diagonal
        const long col_blocks = std::max<long>(1, ((long)n_cols + 255) / 256 / (mode == lcsgpu::MODE_TRIANGLE ? 2 : 1));
        refs_per_wg[b] = lcsgpu::refs_per_block_for(buckets[b].bv, buckets[b].quirk, (long)buckets[b].items.size(), col_blocks);
        is_contig[b] = contiguous(buckets[b]);
        if (is_contig[b] && mode == lcsgpu::MODE_TRIANGLE && buckets[b].bv != 0) {
            // compact grid: only the workgroups at or below the diagonal, bottom row first
            const Bucket& bk = buckets[b];
            const int R = refs_per_wg[b];
            const int nrefs = (int)bk.items.size();
            const int gy = (nrefs + R - 1) / R;
            std::vector<int32_t>& pre = tri_prefix[b];
            pre.resize((size_t)gy + 1);
            int64_t acc = 0;
            for (int k = 0; k < gy; ++k) {
                const int y = gy - 1 - k;
                const int64_t max_row = bk.items[0].row + std::min(nrefs, (y + 1) * R) - 1;
                const int64_t cols = std::min<int64_t>(n_cols, max_row);
                pre[k] = (int32_t)acc;
                acc += cols > 0 ? (cols + 255) / 256 : 0;
            }
            if (acc > 0x7fffffff) return fail(LCSGPU_E_INVALID, "triangle grid too large (%lld workgroups)", (long long)acc);
            pre[gy] = (int32_t)acc;
            pre_off[b] = bytes;
            bytes += align8(pre.size() * 4);
        }
        if (is_contig[b]) continue;
        row_off[b] = bytes;
        bytes += align8(buckets[b].items.size() * 8);
        id_off[b] = bytes;
        bytes += align8(buckets[b].items.size() * 4);
    }
    HIP_TRY(hipSetDevice(ctx->device));
    if (bytes) {
        if (L.plan_in_flight) {
            HIP_TRY(hipStreamSynchronize(L.stream));
            L.plan_in_flight = false;
        }
        HIP_TRY(L.h_plan.reserve(bytes));
        HIP_TRY(L.d_plan.reserve(bytes));
        char* h = (char*)L.h_plan.p;
        if (col_ids) memcpy(h + col_off, col_ids, (size_t)n_cols * 4);
        for (size_t b = 0; b < buckets.size(); ++b) {
            if (!tri_prefix[b].empty()) memcpy(h + pre_off[b], tri_prefix[b].data(), tri_prefix[b].size() * 4);
            if (is_contig[b]) continue;
            int64_t* hr = (int64_t*)(h + row_off[b]);
            int32_t* hi = (int32_t*)(h + id_off[b]);
            for (size_t k = 0; k < buckets[b].items.size(); ++k) {
                hr[k] = buckets[b].items[k].row;
                hi[k] = buckets[b].items[k].id;
            }
        }
        HIP_TRY(hipMemcpyAsync(L.d_plan.p, h, bytes, hipMemcpyHostToDevice, L.stream));
        L.plan_in_flight = true;
    }

    HIP_TRY(hipEventRecord(L.ev_start, L.stream));
    for (size_t b = 0; b < buckets.size(); ++b) {
        const Bucket& bk = buckets[b];
        RowsArgs a{};
        a.tiles = (const uint8_t*)ctx->d_tiles.p;
        a.tile_base = (const uint64_t*)ctx->d_tile_base.p;
        a.lens = (const uint32_t*)ctx->d_lens.p;
        a.n_refs = (int32_t)bk.items.size();
        if (is_contig[b]) {
            a.ref_ids = nullptr;
            a.ref_rows = nullptr;
            a.ref_begin = bk.items[0].id;
            a.row0 = bk.items[0].row;
        } else {
            a.ref_ids = (const int32_t*)((char*)L.d_plan.p + id_off[b]);
            a.ref_rows = (const int64_t*)((char*)L.d_plan.p + row_off[b]);
        }
        a.col_ids = col_ids ? (const int32_t*)((char*)L.d_plan.p + col_off) : nullptr;
        a.col_begin = col_begin;
        a.n_cols = n_cols;
        a.out = d_out;
        a.ld = ld;
        a.out_offset = out_offset;
        a.elem_size = elem_size;
        a.mode = mode;
        a.refs_per_block = refs_per_wg[b];
        int32_t use_cols = n_cols;
        if (mode == lcsgpu::MODE_TRIANGLE) { // columns at or beyond the largest row are never wanted
            int64_t max_row = 0;
            for (const RefItem& it : bk.items) max_row = std::max(max_row, it.row);
            use_cols = (int32_t)std::min<int64_t>(n_cols, max_row);
            if (use_cols <= 0) continue;
        }
        const int gx = (use_cols + 255) / 256;
        const int gy = (a.n_refs + a.refs_per_block - 1) / a.refs_per_block;
        if (bk.bv != 0 && !tri_prefix[b].empty()) {
            a.tri_prefix = (const int32_t*)((char*)L.d_plan.p + pre_off[b]);
            a.tri_rows = (int32_t)tri_prefix[b].size() - 1;
            const int total = tri_prefix[b].back();
            if (total > 0) {
                HIP_TRY(lcsgpu::launch_rows(bk.bv, bk.quirk, a, total, 1, L.stream));
                ++L.last_launches;
            }
        } else if (bk.bv != 0) {
            if (gy > 65535) return fail(LCSGPU_E_INVALID, "too many ref tiles in one call (%d)", gy);
            HIP_TRY(lcsgpu::launch_rows(bk.bv, bk.quirk, a, gx, gy, L.stream));
            ++L.last_launches;
        } else {
            // long refs: slices of ref blocks so the carry scratch stays bounded
            const int n_chunks_max = (int)((ctx->max_len + 15) / 16);
            // ref-tile rows per launch: as many as fit 2 GiB of carry scratch (512 B per chunk per workgroup)
            const size_t per_block = (size_t)n_chunks_max * 512;
            const int gy_step = (int)std::max<size_t>(1, ((size_t)2 << 30) / per_block / (size_t)gx);
            HIP_TRY(L.d_carry.reserve(lcsgpu::long_carry_bytes(gx, std::min(gy, gy_step), n_chunks_max)));
            for (int y0 = 0; y0 < gy; y0 += gy_step) {
                RowsArgs s = a;
                const int first = y0 * a.refs_per_block;
                s.n_refs = std::min(a.n_refs - first, gy_step * a.refs_per_block);
                if (is_contig[b]) {
                    s.ref_begin = a.ref_begin + first;
                    s.row0 = a.row0 + first;
                } else {
                    s.ref_ids = a.ref_ids + first;
                    s.ref_rows = a.ref_rows + first;
                }
                const int sgy = (s.n_refs + s.refs_per_block - 1) / s.refs_per_block;
                HIP_TRY(lcsgpu::launch_long(bk.quirk, s, gx, sgy, L.d_carry.p, n_chunks_max, L.stream));
                ++L.last_launches;
            }
        }
    }
    HIP_TRY(hipEventRecord(L.ev_stop, L.stream));
    L.timing_valid = true;
    return LCSGPU_OK;
}

// After a host-memory call has been synchronised: account its kernel time.
void finish_host_call(lcsgpu_ctx* ctx, Lane& L)
{
    L.plan_in_flight = false;
    float f = 0.f;
    if (L.timing_valid && L.last_launches > 0 && hipEventElapsedTime(&f, L.ev_start, L.ev_stop) != hipSuccess) f = 0.f;
    g_last.ctx = ctx;
    g_last.pending_on_lane0 = false;
    g_last.ms = f;
    g_last.launches = L.last_launches;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->total_kernel_ms += f;
}


// det_uniform_int_distribution<int>(n_medoids, n_elems - 1) over the owner's generator
// (deterministic_random.h:62-76), appended to the job's draws and copied to the device.
int clarans_extend_draws(ClaransJob& j, size_t want, hipStream_t stream)
{
    std::vector<int32_t>& draws = *j.draws;
    if (draws.size() < want) {
        const uint32_t k = (uint32_t)j.a.n_medoids, diff = (uint32_t)(j.a.n_elems - j.a.n_medoids);
        const uint32_t bad = 0xffffffffu / diff;
        const size_t old = draws.size();
        want = std::max(want, old * 2);
        draws.reserve(want);
        while (draws.size() < want) {
            const uint32_t r = (*j.gen_positions)();
            if (r / diff < bad) draws.push_back((int32_t)(r % diff + k));
        }
        const bool regrow = j.d_draws->cap < want * 4;
        HIP_TRY(j.d_draws->reserve(want * 4));
        const size_t from = regrow ? 0 : old;
        HIP_TRY(hipMemcpyAsync((int32_t*)j.d_draws->p + from, draws.data() + from, (draws.size() - from) * 4,
                               hipMemcpyHostToDevice, stream));
    }
    j.a.draws = (const int32_t*)j.d_draws->p;
    j.a.draws_len = (int32_t)draws.size();
    return LCSGPU_OK;
}

// One stint as the driver: rounds for everything joined, until nothing is left or `mine` is done.
void clarans_drive(lcsgpu_ctx* ctx, ClaransBatcher& B, ClaransJob* mine)
{
    const int rounds_per_look = 32;
    for (;;) {
        std::vector<ClaransJob*> now;
        {
            std::lock_guard<std::mutex> lk(B.mu);
            for (ClaransJob* j : B.joined)
                if ((int)now.size() < lcsgpu::CLARANS_MAX_BATCH) now.push_back(j);
            if (now.empty() || mine->done) {
                B.driver_present = false;
                B.cv.notify_all();
                return;
            }
        }
        int rc = LCSGPU_OK;
        const auto t_look = std::chrono::steady_clock::now();
        lcsgpu::ClaransBatch batch{};
        for (ClaransJob* j : now) {
            if (rc == LCSGPU_OK && j->a.n_elems > j->a.n_medoids) // a round uses at most `corrected` draws and prepares the next window
                rc = clarans_extend_draws(*j, (size_t)j->p_host + (size_t)(rounds_per_look + 1) * std::max(j->a.corrected, 1), B.stream);
            batch.s[batch.n++] = j->a;
        }
        auto hip_ok = [&](hipError_t e, const char* what) {
            if (e != hipSuccess && rc == LCSGPU_OK) rc = fail(LCSGPU_E_HIP, "%s failed: %s", what, hipGetErrorString(e));
        };
        if (rc == LCSGPU_OK) hip_ok(lcsgpu::launch_clarans_rounds(batch, rounds_per_look, B.stream), "CLARANS rounds");
        int32_t* hs = (int32_t*)B.h_states.p;
        for (size_t i = 0; i < now.size() && rc == LCSGPU_OK; ++i)
            hip_ok(hipMemcpyAsync(hs + 16 * i, now[i]->a.state, 64, hipMemcpyDeviceToHost, B.stream), "state read-back");
        if (rc == LCSGPU_OK) hip_ok(hipEventRecord(B.ev, B.stream), "hipEventRecord");
        if (rc == LCSGPU_OK) hip_ok(hipEventSynchronize(B.ev), "hipEventSynchronize");
        else (void)hipStreamSynchronize(B.stream);
        {
            std::lock_guard<std::mutex> lk(B.mu);
            const std::string msg = rc == LCSGPU_OK ? std::string() : std::string(lcsgpu_last_error());
            B.prof_looks[now.size()]++;
            B.prof_seconds[now.size()] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_look).count();
            for (size_t i = 0; i < now.size(); ++i) {
                ClaransJob* j = now[i];
                if (rc == LCSGPU_OK) {
                    memcpy(j->state, hs + 16 * i, 64);
                    j->p_host = j->state[0];
                    if (j->state[6]) {
                        j->rc = LCSGPU_E_STATE;
                        j->error = "CLARANS: the device search ran out of pre-drawn steps";
                    }
                } else {
                    j->rc = rc;
                    j->error = msg;
                }
                if (j->rc != LCSGPU_OK || j->state[1]) {
                    j->done = true;
                    B.joined.erase(std::find(B.joined.begin(), B.joined.end(), j));
                }
            }
            B.cv.notify_all();
        }
    }
}

// Join the batch with a search whose device state is initialised; returns when it has finished.
int clarans_run_search(lcsgpu_ctx* ctx, ClaransJob& job)
{
    ClaransBatcher& B = ctx->clarans_groups[ctx->clarans_next++ % ctx->clarans_groups.size()];
    job.done = false;
    std::unique_lock<std::mutex> lk(B.mu);
    B.joined.push_back(&job);
    while (!job.done) {
        if (!B.driver_present) {
            B.driver_present = true;
            lk.unlock();
            clarans_drive(ctx, B, &job);
            lk.lock();
        } else {
            B.cv.wait(lk, [&] { return job.done || !B.driver_present; });
        }
    }
    lk.unlock();
    if (job.rc != LCSGPU_OK) return fail(job.rc, "%s", job.error.c_str());
    return LCSGPU_OK;
}

void note_async_call(lcsgpu_ctx* ctx)
{
    g_last.ctx = ctx;
    g_last.pending_on_lane0 = true;
}

} // namespace

extern "C" {

const char* lcsgpu_version(void) { return "lcsgpu 0.1 gfx950"; }
const char* lcsgpu_last_error(void) { return g_err.c_str(); }

int lcsgpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int lcsgpu_create(int device_id, lcsgpu_ctx** out_ctx)
{
    if (!out_ctx) return fail(LCSGPU_E_INVALID, "out_ctx is NULL");
    *out_ctx = nullptr;
    int n_lanes = 16;
    if (const char* e = getenv("LCSGPU_LANES")) n_lanes = std::max(1, std::min(64, atoi(e)));
    // One hardware queue per lane, so that the lanes' small kernels really overlap: the runtime's
    // default of 4 queues makes 16 streams share them and serialises their launches (1 M-sequence
    // MedoidTree: 4.7 s -> 2.6 s of tree stage).  Read by the HIP runtime when it initialises, i.e.
    // effective if this is the process's first HIP call; an explicit setting by the user wins.
    {
        char buf[16];
        snprintf(buf, sizeof buf, "%d", n_lanes);
        setenv("GPU_MAX_HW_QUEUES", buf, 0);
    }
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(LCSGPU_E_NODEVICE, "no HIP device available (%s)", hipGetErrorString(e));
    if (device_id < 0 || device_id >= n)
        return fail(LCSGPU_E_INVALID, "device %d not in [0,%d)", device_id, n);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(LCSGPU_E_NODEVICE, "device %d is %s; this library is built for gfx950 only", device_id,
                    prop.gcnArchName);
    HIP_TRY(hipSetDevice(device_id));
    lcsgpu_ctx* ctx = new (std::nothrow) lcsgpu_ctx;
    if (!ctx) return fail(LCSGPU_E_NOMEM, "out of host memory");
    ctx->device = device_id;
    ctx->lanes.resize(n_lanes);
    for (Lane& l : ctx->lanes)
        if (hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreate(&l.ev_start) != hipSuccess || hipEventCreate(&l.ev_stop) != hipSuccess ||
            hipEventCreateWithFlags(&l.ev_done, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess) {
            lcsgpu_destroy(ctx);
            return fail(LCSGPU_E_HIP, "stream/event creation failed");
        }
    int n_groups = 4; // 3 x 10^6-sequence MedoidTree, tree stage: 1 group 2.92 s, 2: 2.79 s, 4: 2.71 s, 8: 4.13 s
    if (const char* e = getenv("LCSGPU_CLARANS_GROUPS")) n_groups = std::max(1, std::min(16, atoi(e)));
    ctx->clarans_groups = std::vector<ClaransBatcher>(n_groups);
    for (ClaransBatcher& B : ctx->clarans_groups)
        if (hipStreamCreateWithFlags(&B.stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&B.ev, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess ||
            B.h_states.reserve(lcsgpu::CLARANS_MAX_BATCH * 64) != hipSuccess) {
            lcsgpu_destroy(ctx);
            return fail(LCSGPU_E_HIP, "stream/event creation failed");
        }
    *out_ctx = ctx;
    return LCSGPU_OK;
}

int lcsgpu_destroy(lcsgpu_ctx* ctx)
{
    if (!ctx) return LCSGPU_OK;
    (void)hipSetDevice(ctx->device);
    for (Lane& l : ctx->lanes) {
        if (l.stream) (void)hipStreamSynchronize(l.stream);
        l.d_plan.release();
        l.d_out.release();
        l.d_carry.release();
        l.d_work.release();
        l.d_draws.release();
        l.h_plan.release();
        l.h_small.release();
        if (l.ev_start) (void)hipEventDestroy(l.ev_start);
        if (l.ev_stop) (void)hipEventDestroy(l.ev_stop);
        if (l.ev_done) (void)hipEventDestroy(l.ev_done);
        if (l.stream) (void)hipStreamDestroy(l.stream);
    }
    for (ClaransBatcher& B : ctx->clarans_groups) {
        if (getenv("LCSGPU_PROFILE"))
            for (int i = 1; i <= lcsgpu::CLARANS_MAX_BATCH; ++i)
                if (B.prof_looks[i])
                    fprintf(stderr, "clarans.batch[%d searches]: %ld looks of 32 rounds, %.3f s, %.1f us per round\n", i,
                            B.prof_looks[i], B.prof_seconds[i], 1e6 * B.prof_seconds[i] / B.prof_looks[i] / 32);
        if (B.stream) { (void)hipStreamSynchronize(B.stream); (void)hipStreamDestroy(B.stream); }
        if (B.ev) (void)hipEventDestroy(B.ev);
        B.h_states.release();
    }
    ctx->d_tiles.release();
    ctx->d_tile_base.release();
    ctx->d_lens.release();
    ctx->d_pow.release();
    ctx->d_powf.release();
    ctx->d_prim.release();
    ctx->d_qrows.release();
    ctx->d_qcols.release();
    ctx->d_dist.release();
    delete ctx;
    return LCSGPU_OK;
}

int lcsgpu_encode(const char* residues, size_t n, uint8_t* codes, size_t* n_codes)
{
    if ((!residues && n) || !codes || !n_codes) return fail(LCSGPU_E_INVALID, "NULL argument");
    // The alphabet and the folding of characters above 'Z' of CSequence's constructor
    // (reference core/sequence.cpp:17,53-79); the lookup covers the terminator too.
    static const char alphabet[25] = "ARNDCQEGHILKMFPSTWYVBZX*";
    uint8_t lut[256];
    for (int ch = 0; ch < 256; ++ch) {
        char c = (char)ch;
        if (c > 'Z') c = (char)(c - 32);
        uint8_t code = 22;
        for (int i = 0; i < 25; ++i)
            if (alphabet[i] == c) {
                code = (uint8_t)i;
                break;
            }
        lut[ch] = code;
    }
    size_t m = 0;
    for (size_t i = 0; i < n; ++i)
        if (residues[i] != '-') codes[m++] = lut[(unsigned char)residues[i]];
    *n_codes = m;
    return LCSGPU_OK;
}

int lcsgpu_upload(lcsgpu_ctx* ctx, const uint8_t* codes, const uint64_t* offsets, int32_t n)
{
    if (!ctx || !offsets || n < 0 || (!codes && n > 0 && offsets[n] > 0))
        return fail(LCSGPU_E_INVALID, "bad argument");
    LaneGuard guard(ctx, LaneGuard::ALL); // nothing may run while the set is replaced
    HIP_TRY(hipSetDevice(ctx->device));
    for (Lane& l : ctx->lanes) HIP_TRY(hipStreamSynchronize(l.stream));
    ctx->n = -1;
    std::vector<uint32_t> lens(n);
    std::vector<uint8_t> quirk(n);
    uint32_t max_len = 0;
    for (int32_t i = 0; i < n; ++i) {
        if (offsets[i + 1] < offsets[i]) return fail(LCSGPU_E_INVALID, "offsets not monotone at %d", i);
        const uint64_t len = offsets[i + 1] - offsets[i];
        if (len > 0x7fffffffu) return fail(LCSGPU_E_INVALID, "sequence %d too long", i);
        lens[i] = (uint32_t)len;
        max_len = std::max(max_len, lens[i]);
    }
    // position-major tiles of 64 sequences; chunk = 16 residues; byte = code*8; pad = 22*8.
    // The packed codes go to HBM as they are; tiles and orientation flags are built there.
    const int32_t n_tiles = (n + 63) / 64;
    std::vector<uint64_t> tile_base((size_t)n_tiles + 1, 0);
    for (int32_t t = 0; t < n_tiles; ++t) {
        uint32_t tmax = 0;
        for (int32_t s = t * 64; s < std::min(n, (t + 1) * 64); ++s) tmax = std::max(tmax, lens[s]);
        const uint64_t chunks = std::max<uint64_t>(1, (tmax + 15) / 16);
        tile_base[t + 1] = tile_base[t] + chunks * 1024;
    }
    const size_t total = (size_t)tile_base[n_tiles];
    const size_t raw_bytes = n ? (size_t)(offsets[n] - offsets[0]) : 0;
    if (n && offsets[0] != 0) return fail(LCSGPU_E_INVALID, "offsets[0] must be 0");
    hipError_t e = hipSuccess;
    if ((e = ctx->d_tiles.reserve(std::max<size_t>(total, 16))) != hipSuccess ||
        (e = ctx->d_tile_base.reserve(((size_t)n_tiles + 1) * 8)) != hipSuccess ||
        (e = ctx->d_lens.reserve(std::max<size_t>((size_t)n * 4, 16))) != hipSuccess)
        return fail(LCSGPU_E_NOMEM, "device allocation failed: %s", hipGetErrorString(e));
    HIP_TRY(hipMemcpy(ctx->d_tile_base.p, tile_base.data(), ((size_t)n_tiles + 1) * 8, hipMemcpyHostToDevice));
    if (n) HIP_TRY(hipMemcpy(ctx->d_lens.p, lens.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    if (n) {
        DevBuf d_raw, d_off, d_quirk; // only needed while the tiles are built
        struct Release {
            DevBuf &a, &b, &c;
            ~Release() { a.release(); b.release(); c.release(); }
        } release{d_raw, d_off, d_quirk};
        if ((e = d_raw.reserve(std::max<size_t>(raw_bytes, 16))) != hipSuccess ||
            (e = d_off.reserve(((size_t)n + 1) * 8)) != hipSuccess || (e = d_quirk.reserve((size_t)n + 16)) != hipSuccess)
            return fail(LCSGPU_E_NOMEM, "device allocation failed: %s", hipGetErrorString(e));
        hipStream_t st = ctx->lanes[0].stream;
        if (raw_bytes) HIP_TRY(hipMemcpyAsync(d_raw.p, codes, raw_bytes, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_off.p, offsets, ((size_t)n + 1) * 8, hipMemcpyHostToDevice, st));
        int32_t* d_flags = (int32_t*)((char*)d_quirk.p + (((size_t)n + 3) & ~(size_t)3));
        HIP_TRY(hipMemsetAsync(d_flags, 0, 4, st));
        HIP_TRY(lcsgpu::launch_build_set((const uint8_t*)d_raw.p, (const uint64_t*)d_off.p,
                                         (const uint64_t*)ctx->d_tile_base.p, n, (uint8_t*)ctx->d_tiles.p,
                                         (uint8_t*)d_quirk.p, d_flags, st));
        int32_t flags = 0;
        HIP_TRY(hipMemcpyAsync(quirk.data(), d_quirk.p, (size_t)n, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(&flags, d_flags, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (flags & 1) return fail(LCSGPU_E_INVALID, "symbol code out of range (>= 32) in the uploaded set");
    }
    {
        // pow(indel, 0.75) for every possible indel, from the host's libm -- the entries of
        // Transform<double, indel075_div_lcs>::pp_pow075_rec (reference AbstractTreeGenerator.hpp:43-48)
        std::vector<double> pw((size_t)2 * max_len + 1);
        for (size_t i = 0; i < pw.size(); ++i) pw[i] = pow((double)(uint32_t)i, 0.75);
        HIP_TRY(ctx->d_pow.reserve(pw.size() * 8));
        HIP_TRY(hipMemcpy(ctx->d_pow.p, pw.data(), pw.size() * 8, hipMemcpyHostToDevice));
        // the float table of Transform<float, indel075_div_lcs>: (float) pow((double) i, 0.75)
        std::vector<float> pf(pw.size());
        for (size_t i = 0; i < pw.size(); ++i) pf[i] = (float)pw[i];
        HIP_TRY(ctx->d_powf.reserve(pf.size() * 4));
        HIP_TRY(hipMemcpy(ctx->d_powf.p, pf.data(), pf.size() * 4, hipMemcpyHostToDevice));
    }
    ctx->lens.swap(lens);
    ctx->quirk.swap(quirk);
    ctx->max_len = max_len;
    ctx->n = n;
    return LCSGPU_OK;
}

int32_t lcsgpu_count(lcsgpu_ctx* ctx)
{
    if (!ctx) return fail(LCSGPU_E_INVALID, "NULL ctx");
    return ctx->n;
}

int32_t lcsgpu_length(lcsgpu_ctx* ctx, int32_t i)
{
    if (!ctx || i < 0 || i >= ctx->n) return fail(LCSGPU_E_INVALID, "bad index");
    return (int32_t)ctx->lens[i];
}

int32_t lcsgpu_orientation_flags(lcsgpu_ctx* ctx, uint8_t* flags)
{
    if (!ctx) return fail(LCSGPU_E_INVALID, "NULL ctx");
    if (ctx->n < 0) return fail(LCSGPU_E_STATE, "no sequence set uploaded");
    int32_t count = 0;
    for (int32_t i = 0; i < ctx->n; ++i) {
        if (flags) flags[i] = ctx->quirk[i];
        count += ctx->quirk[i] ? 1 : 0;
    }
    return count;
}

int lcsgpu_lcs_rect_dev(lcsgpu_ctx* ctx, const int32_t* ref_ids, int32_t ref_begin, int32_t n_refs,
                        const int32_t* col_ids, int32_t col_begin, int32_t n_cols, void* d_out,
                        int64_t ld, int elem_size, int sync)
{
    if (!ctx || (!d_out && n_refs > 0 && n_cols > 0)) return fail(LCSGPU_E_INVALID, "NULL argument");
    if (ld < n_cols) return fail(LCSGPU_E_INVALID, "ld < n_cols");
    LaneGuard guard(ctx, LaneGuard::LANE0);
    Lane& L = guard.lane();
    int rc = run_rows(ctx, L, lcsgpu::MODE_RECT, ref_ids, ref_begin, n_refs, col_ids, col_begin, n_cols, d_out, ld, 0,
                      elem_size);
    if (rc) return rc;
    note_async_call(ctx);
    if (sync) HIP_TRY(hipStreamSynchronize(L.stream));
    return LCSGPU_OK;
}

int lcsgpu_lcs_rect(lcsgpu_ctx* ctx, const int32_t* ref_ids, int32_t ref_begin, int32_t n_refs,
                    const int32_t* col_ids, int32_t col_begin, int32_t n_cols, void* out, int64_t ld,
                    int elem_size)
{
    if (!ctx || (!out && n_refs > 0 && n_cols > 0)) return fail(LCSGPU_E_INVALID, "NULL argument");
    if (ld < n_cols) return fail(LCSGPU_E_INVALID, "ld < n_cols");
    if (n_refs <= 0 || n_cols <= 0) return (n_refs < 0 || n_cols < 0) ? fail(LCSGPU_E_INVALID, "negative count") : LCSGPU_OK;
    if (elem_size != 2 && elem_size != 4) return fail(LCSGPU_E_INVALID, "elem_size must be 2 or 4");
    LaneGuard guard(ctx, LaneGuard::ANY);
    Lane& L = guard.lane();
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t bytes = (size_t)n_refs * n_cols * elem_size;
    HIP_TRY(L.d_out.reserve(bytes));
    int rc = run_rows(ctx, L, lcsgpu::MODE_RECT, ref_ids, ref_begin, n_refs, col_ids, col_begin, n_cols, L.d_out.p,
                      n_cols, 0, elem_size);
    if (rc) return rc;
    HIP_TRY(hipMemcpy2DAsync(out, (size_t)ld * elem_size, L.d_out.p, (size_t)n_cols * elem_size,
                             (size_t)n_cols * elem_size, n_refs, hipMemcpyDeviceToHost, L.stream));
    HIP_TRY(hipEventRecord(L.ev_done, L.stream));
    HIP_TRY(hipEventSynchronize(L.ev_done)); // sleeps; worker threads must not burn a core per pending call
    finish_host_call(ctx, L);
    return LCSGPU_OK;
}

static int triangle_common(lcsgpu_ctx* ctx, Lane& L, int32_t row_begin, int32_t row_end, void* d_out, int elem_size)
{
    const int64_t off = (int64_t)row_begin * (row_begin - 1) / 2;
    return run_rows(ctx, L, lcsgpu::MODE_TRIANGLE, nullptr, row_begin, row_end - row_begin, nullptr, 0,
                    std::max(0, row_end - 1), d_out, 0, off, elem_size, row_begin);
}

int lcsgpu_lcs_triangle_dev(lcsgpu_ctx* ctx, int32_t row_begin, int32_t row_end, void* d_out, int elem_size,
                            int sync)
{
    if (!ctx) return fail(LCSGPU_E_INVALID, "NULL ctx");
    if (ctx->n < 0) return fail(LCSGPU_E_STATE, "no sequence set uploaded");
    if (row_begin < 0 || row_end < row_begin || row_end > ctx->n) return fail(LCSGPU_E_INVALID, "bad row range");
    LaneGuard guard(ctx, LaneGuard::LANE0);
    Lane& L = guard.lane();
    int rc = triangle_common(ctx, L, row_begin, row_end, d_out, elem_size);
    if (rc) return rc;
    note_async_call(ctx);
    if (sync) HIP_TRY(hipStreamSynchronize(L.stream));
    return LCSGPU_OK;
}

int lcsgpu_lcs_triangle(lcsgpu_ctx* ctx, int32_t row_begin, int32_t row_end, void* out, int elem_size)
{
    if (!ctx) return fail(LCSGPU_E_INVALID, "NULL ctx");
    if (ctx->n < 0) return fail(LCSGPU_E_STATE, "no sequence set uploaded");
    if (row_begin < 0 || row_end < row_begin || row_end > ctx->n) return fail(LCSGPU_E_INVALID, "bad row range");
    if (elem_size != 2 && elem_size != 4) return fail(LCSGPU_E_INVALID, "elem_size must be 2 or 4");
    const int64_t count = (int64_t)row_end * (row_end - 1) / 2 - (int64_t)row_begin * (row_begin - 1) / 2;
    if (count <= 0) return LCSGPU_OK;
    if (!out) return fail(LCSGPU_E_INVALID, "NULL out");
    LaneGuard guard(ctx, LaneGuard::ANY);
    Lane& L = guard.lane();
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(L.d_out.reserve((size_t)count * elem_size));
    int rc = triangle_common(ctx, L, row_begin, row_end, L.d_out.p, elem_size);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, L.d_out.p, (size_t)count * elem_size, hipMemcpyDeviceToHost, L.stream));
    HIP_TRY(hipEventRecord(L.ev_done, L.stream));
    HIP_TRY(hipEventSynchronize(L.ev_done)); // sleeps; worker threads must not burn a core per pending call
    finish_host_call(ctx, L);
    return LCSGPU_OK;
}

int lcsgpu_row_minima_dev(lcsgpu_ctx* ctx, const void* d_triangle, int elem_size, int32_t row_begin,
                          int32_t row_end, int distance_kind, void* d_out, int sync)
{
    if (!ctx) return fail(LCSGPU_E_INVALID, "NULL ctx");
    if (ctx->n < 0) return fail(LCSGPU_E_STATE, "no sequence set uploaded");
    if (row_begin < 0 || row_end < row_begin || row_end > ctx->n) return fail(LCSGPU_E_INVALID, "bad row range");
    if (elem_size != 2 && elem_size != 4) return fail(LCSGPU_E_INVALID, "elem_size must be 2 or 4");
    if (distance_kind != LCSGPU_DIST_INDEL_DIV_LCS && distance_kind != LCSGPU_DIST_INDEL075_DIV_LCS)
        return fail(LCSGPU_E_INVALID, "unknown distance kind %d", distance_kind);
    if (row_end == row_begin) return LCSGPU_OK;
    if (!d_triangle || !d_out) return fail(LCSGPU_E_INVALID, "NULL device pointer");
    LaneGuard guard(ctx, LaneGuard::LANE0);
    Lane& L = guard.lane();
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(lcsgpu::launch_row_minima(d_triangle, elem_size, row_begin, row_end, (const uint32_t*)ctx->d_lens.p,
                                      (const double*)ctx->d_pow.p, distance_kind, (lcsgpu::RowMin*)d_out, L.stream));
    if (sync) HIP_TRY(hipStreamSynchronize(L.stream));
    return LCSGPU_OK;
}

int lcsgpu_mst_prim(lcsgpu_ctx* ctx, int distance_kind, lcsgpu_mst_edge* out_edges)
{
    const bool triangle_orientation = (distance_kind & LCSGPU_MST_TRIANGLE_ORIENTATION) != 0;
    distance_kind &= ~LCSGPU_MST_TRIANGLE_ORIENTATION;
    if (!ctx) return fail(LCSGPU_E_INVALID, "NULL ctx");
    if (ctx->n < 0) return fail(LCSGPU_E_STATE, "no sequence set uploaded");
    if (distance_kind != LCSGPU_DIST_INDEL_DIV_LCS && distance_kind != LCSGPU_DIST_INDEL075_DIV_LCS)
        return fail(LCSGPU_E_INVALID, "unknown distance kind %d", distance_kind);
    const int32_t n = ctx->n;
    if (n < 2) return LCSGPU_OK;
    if (!out_edges) return fail(LCSGPU_E_INVALID, "NULL out_edges");
    LaneGuard guard(ctx, LaneGuard::LANE0);
    Lane& L = guard.lane();
    HIP_TRY(hipSetDevice(ctx->device));
    const int elem = ctx->max_len > 65535 ? 4 : 2;
    const size_t pairs = (size_t)n * (n - 1) / 2;
    HIP_TRY(L.d_out.reserve(pairs * elem));
    int rc = run_rows(ctx, L, lcsgpu::MODE_TRIANGLE, nullptr, 0, n, nullptr, 0, n - 1, L.d_out.p, 0, 0, elem);
    if (rc) return rc;

    // orientation-sensitive sequences: their values in both roles, as side tables
    std::vector<int32_t> qindex(n, -1), qlist;
    for (int32_t i = 0; i < n && !triangle_orientation; ++i)
        if (ctx->quirk[i]) {
            qindex[i] = (int32_t)qlist.size();
            qlist.push_back(i);
        }
    const int32_t nq = (int32_t)qlist.size();
    if (nq) {
        HIP_TRY(ctx->d_qrows.reserve((size_t)nq * n * 4));
        HIP_TRY(ctx->d_qcols.reserve((size_t)nq * n * 4));
        rc = run_rows(ctx, L, lcsgpu::MODE_RECT, qlist.data(), 0, nq, nullptr, 0, n, ctx->d_qrows.p, n, 0, 4);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(L.stream)); // the staging buffer of the plan is reused by the next call
        L.plan_in_flight = false;
        rc = run_rows(ctx, L, lcsgpu::MODE_RECT, nullptr, 0, n, qlist.data(), 0, nq, ctx->d_qcols.p, nq, 0, 4);
        if (rc) return rc;
    }

    const int blocks = (n + 255) / 256;
    auto a8 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t o_keyd = 0, o_keyi = o_keyd + a8((size_t)n * 8), o_proc = o_keyi + a8((size_t)n * 8),
                 o_part = o_proc + a8((size_t)n), o_edges = o_part + a8((size_t)2 * blocks * sizeof(lcsgpu::PrimPartial)),
                 o_qidx = o_edges + a8((size_t)(n - 1) * sizeof(lcsgpu::MstEdge)), total = o_qidx + a8((size_t)n * 4);
    HIP_TRY(ctx->d_prim.reserve(total));
    char* base = (char*)ctx->d_prim.p;
    if (nq) HIP_TRY(hipMemcpyAsync(base + o_qidx, qindex.data(), (size_t)n * 4, hipMemcpyHostToDevice, L.stream));
    lcsgpu::PrimArgs a{};
    a.tri = L.d_out.p;
    a.lens = (const uint32_t*)ctx->d_lens.p;
    a.pow_table = (const double*)ctx->d_pow.p;
    a.qindex = nq ? (const int32_t*)(base + o_qidx) : nullptr;
    a.q_rows = (const uint32_t*)ctx->d_qrows.p;
    a.q_cols = (const uint32_t*)ctx->d_qcols.p;
    a.n_q = nq;
    a.n = n;
    a.kind = distance_kind;
    a.n_blocks = blocks;
    a.key_d = (double*)(base + o_keyd);
    a.key_id = (uint64_t*)(base + o_keyi);
    a.processed = (uint8_t*)(base + o_proc);
    a.partials = (lcsgpu::PrimPartial*)(base + o_part);
    a.edges = (lcsgpu::MstEdge*)(base + o_edges);
    HIP_TRY(lcsgpu::launch_prim(a, elem, L.stream));
    static_assert(sizeof(lcsgpu_mst_edge) == sizeof(lcsgpu::MstEdge), "edge layout");
    HIP_TRY(hipMemcpyAsync(out_edges, a.edges, (size_t)(n - 1) * sizeof(lcsgpu_mst_edge), hipMemcpyDeviceToHost,
                           L.stream));
    HIP_TRY(hipStreamSynchronize(L.stream));
    L.plan_in_flight = false;
    note_async_call(ctx);
    return LCSGPU_OK;
}

int lcsgpu_upgma(lcsgpu_ctx* ctx, int distance_kind, int modified, int32_t* out_left, int32_t* out_right)
{
    if (!ctx) return fail(LCSGPU_E_INVALID, "NULL ctx");
    if (ctx->n < 0) return fail(LCSGPU_E_STATE, "no sequence set uploaded");
    if (distance_kind != LCSGPU_DIST_INDEL_DIV_LCS && distance_kind != LCSGPU_DIST_INDEL075_DIV_LCS)
        return fail(LCSGPU_E_INVALID, "unknown distance kind %d", distance_kind);
    const int32_t n = ctx->n;
    if (n < 2) return LCSGPU_OK;
    if (!out_left || !out_right) return fail(LCSGPU_E_INVALID, "NULL output");
    LaneGuard guard(ctx, LaneGuard::LANE0);
    Lane& L = guard.lane();
    HIP_TRY(hipSetDevice(ctx->device));
    const int elem = ctx->max_len > 65535 ? 4 : 2;
    const size_t pairs = (size_t)n * (n - 1) / 2;
    HIP_TRY(L.d_out.reserve(pairs * elem));
    HIP_TRY(ctx->d_dist.reserve(pairs * sizeof(float)));
    int rc = run_rows(ctx, L, lcsgpu::MODE_TRIANGLE, nullptr, 0, n, nullptr, 0, n - 1, L.d_out.p, 0, 0, elem);
    if (rc) return rc;
    const int blocks = (n + 255) / 256;
    auto a16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t o_min = 0, o_near = o_min + a16((size_t)n * 4), o_node = o_near + a16((size_t)n * 4),
                 o_pd = o_node + a16((size_t)n * 4), o_pj = o_pd + a16((size_t)blocks * 4),
                 o_sel = o_pj + a16((size_t)blocks * 4), o_left = o_sel + 16, o_right = o_left + a16((size_t)n * 4),
                 total = o_right + a16((size_t)n * 4);
    HIP_TRY(ctx->d_prim.reserve(total));
    char* base = (char*)ctx->d_prim.p;
    HIP_TRY(hipMemsetAsync(base + o_sel, 0, 16, L.stream));
    lcsgpu::UpgmaArgs a{};
    a.D = (float*)ctx->d_dist.p;
    a.min_dist = (float*)(base + o_min);
    a.nearest = (uint32_t*)(base + o_near);
    a.node_index = (uint32_t*)(base + o_node);
    a.part_d = (float*)(base + o_pd);
    a.part_j = (uint32_t*)(base + o_pj);
    a.sel = (uint32_t*)(base + o_sel);
    a.left = (int32_t*)(base + o_left);
    a.right = (int32_t*)(base + o_right);
    a.n = n;
    a.n_blocks = blocks;
    HIP_TRY(lcsgpu::launch_upgma(a, L.d_out.p, elem, (const uint32_t*)ctx->d_lens.p, (const float*)ctx->d_powf.p,
                                 distance_kind, modified != 0, L.stream));
    uint32_t sel[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(out_left, a.left, (size_t)(n - 1) * 4, hipMemcpyDeviceToHost, L.stream));
    HIP_TRY(hipMemcpyAsync(out_right, a.right, (size_t)(n - 1) * 4, hipMemcpyDeviceToHost, L.stream));
    HIP_TRY(hipMemcpyAsync(sel, a.sel, 16, hipMemcpyDeviceToHost, L.stream));
    HIP_TRY(hipStreamSynchronize(L.stream));
    L.plan_in_flight = false;
    if (sel[2])
        return fail(LCSGPU_E_INVALID, "UPGMA: no finite nearest neighbour (a pair with LCS 0?) -- the reference's "
                                      "algorithm is undefined for this input");
    note_async_call(ctx);
    return LCSGPU_OK;
}

int lcsgpu_nj(lcsgpu_ctx* ctx, int distance_kind, int32_t* out_left, int32_t* out_right)
{
    if (!ctx) return fail(LCSGPU_E_INVALID, "NULL ctx");
    if (ctx->n < 0) return fail(LCSGPU_E_STATE, "no sequence set uploaded");
    if (distance_kind != LCSGPU_DIST_INDEL_DIV_LCS && distance_kind != LCSGPU_DIST_INDEL075_DIV_LCS)
        return fail(LCSGPU_E_INVALID, "unknown distance kind %d", distance_kind);
    const int32_t n = ctx->n;
    if (n < 2) return LCSGPU_OK;
    if (!out_left || !out_right) return fail(LCSGPU_E_INVALID, "NULL output");
    LaneGuard guard(ctx, LaneGuard::LANE0);
    Lane& L = guard.lane();
    HIP_TRY(hipSetDevice(ctx->device));
    const int elem = ctx->max_len > 65535 ? 4 : 2;
    const size_t pairs = (size_t)n * (n - 1) / 2;
    HIP_TRY(L.d_out.reserve(pairs * elem));
    HIP_TRY(ctx->d_dist.reserve(pairs * sizeof(float)));
    int rc = run_rows(ctx, L, lcsgpu::MODE_TRIANGLE, nullptr, 0, n, nullptr, 0, n - 1, L.d_out.p, 0, 0, elem);
    if (rc) return rc;
    auto a16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t o_sum = 0, o_tmp = o_sum + a16((size_t)n * 4), o_pq = o_tmp + a16((size_t)n * 4),
                 o_pi = o_pq + a16((size_t)n * 4), o_node = o_pi + a16((size_t)n * 4), o_act = o_node + a16((size_t)n * 4),
                 o_sel = o_act + a16((size_t)n), o_left = o_sel + 16, o_right = o_left + a16((size_t)n * 4),
                 total = o_right + a16((size_t)n * 4);
    HIP_TRY(ctx->d_prim.reserve(total));
    char* base = (char*)ctx->d_prim.p;
    HIP_TRY(hipMemsetAsync(base + o_sel, 0, 16, L.stream));
    lcsgpu::NjArgs a{};
    a.D = (float*)ctx->d_dist.p;
    a.sum = (float*)(base + o_sum);
    a.tmp = (float*)(base + o_tmp);
    a.part_q = (float*)(base + o_pq);
    a.part_i = (int32_t*)(base + o_pi);
    a.node = (int32_t*)(base + o_node);
    a.active = (uint8_t*)(base + o_act);
    a.sel = (int32_t*)(base + o_sel);
    a.left = (int32_t*)(base + o_left);
    a.right = (int32_t*)(base + o_right);
    a.n = n;
    HIP_TRY(lcsgpu::launch_float_distances(L.d_out.p, elem, (const uint32_t*)ctx->d_lens.p, (const float*)ctx->d_powf.p,
                                           distance_kind, n, a.D, L.stream));
    HIP_TRY(lcsgpu::launch_nj(a, L.stream));
    int32_t sel[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(out_left, a.left, (size_t)(n - 1) * 4, hipMemcpyDeviceToHost, L.stream));
    HIP_TRY(hipMemcpyAsync(out_right, a.right, (size_t)(n - 1) * 4, hipMemcpyDeviceToHost, L.stream));
    HIP_TRY(hipMemcpyAsync(sel, a.sel, 16, hipMemcpyDeviceToHost, L.stream));
    HIP_TRY(hipStreamSynchronize(L.stream));
    L.plan_in_flight = false;
    note_async_call(ctx);
    if (sel[2])
        return fail(LCSGPU_E_INVALID, "NJ: no finite q (a pair with LCS 0?) -- the reference's result is degenerate "
                                      "for this input");
    return LCSGPU_OK;
}

int lcsgpu_lcs_triangle_ids(lcsgpu_ctx* ctx, const int32_t* ids, int32_t n_ids, void* out, int elem_size)
{
    if (!ctx) return fail(LCSGPU_E_INVALID, "NULL ctx");
    if (ctx->n < 0) return fail(LCSGPU_E_STATE, "no sequence set uploaded");
    if (n_ids < 0 || (n_ids > 0 && !ids)) return fail(LCSGPU_E_INVALID, "bad id list");
    if (elem_size != 2 && elem_size != 4) return fail(LCSGPU_E_INVALID, "elem_size must be 2 or 4");
    const int64_t count = (int64_t)n_ids * (n_ids - 1) / 2;
    if (count <= 0) return LCSGPU_OK;
    if (!out) return fail(LCSGPU_E_INVALID, "NULL out");
    LaneGuard guard(ctx, LaneGuard::ANY);
    Lane& L = guard.lane();
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(L.d_out.reserve((size_t)count * elem_size));
    // row k = ids[k] as the ref, column c = ids[c] as the partner, c < k
    int rc = run_rows(ctx, L, lcsgpu::MODE_TRIANGLE, ids, 0, n_ids, ids, 0, n_ids - 1, L.d_out.p, 0, 0, elem_size, 0);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, L.d_out.p, (size_t)count * elem_size, hipMemcpyDeviceToHost, L.stream));
    HIP_TRY(hipEventRecord(L.ev_done, L.stream));
    HIP_TRY(hipEventSynchronize(L.ev_done)); // sleeps; worker threads must not burn a core per pending call
    finish_host_call(ctx, L);
    return LCSGPU_OK;
}

int lcsgpu_lcs_triangles_batch(lcsgpu_ctx* ctx, const int32_t* ids, const int64_t* group_offsets, int32_t n_groups,
                               void* out, int elem_size)
{
    if (!ctx) return fail(LCSGPU_E_INVALID, "NULL ctx");
    if (ctx->n < 0) return fail(LCSGPU_E_STATE, "no sequence set uploaded");
    if (n_groups < 0 || (n_groups > 0 && !group_offsets)) return fail(LCSGPU_E_INVALID, "bad group table");
    if (elem_size != 2 && elem_size != 4) return fail(LCSGPU_E_INVALID, "elem_size must be 2 or 4");
    if (elem_size == 2 && ctx->max_len > 65535)
        return fail(LCSGPU_E_INVALID, "uint16 output needs all sequences <= 65535 residues");
    if (n_groups == 0) return LCSGPU_OK;
    if (group_offsets[0] != 0) return fail(LCSGPU_E_INVALID, "group_offsets[0] must be 0");
    const int64_t n_total = group_offsets[n_groups];
    if (n_total < 0 || n_total > 0x7fffffff) return fail(LCSGPU_E_INVALID, "bad total id count");
    std::vector<int64_t> tri_base((size_t)n_groups + 1, 0);
    bool any_long = false;
    for (int32_t g = 0; g < n_groups; ++g) {
        const int64_t m = group_offsets[g + 1] - group_offsets[g];
        if (m < 0) return fail(LCSGPU_E_INVALID, "group_offsets not ascending");
        tri_base[g + 1] = tri_base[g] + m * (m - 1) / 2;
    }
    const int64_t count = tri_base[n_groups];
    if (count <= 0) return LCSGPU_OK;
    if (!ids || !out) return fail(LCSGPU_E_INVALID, "NULL ids / out");
    for (int64_t p = 0; p < n_total; ++p) {
        if (ids[p] < 0 || ids[p] >= ctx->n) return fail(LCSGPU_E_INVALID, "id %d out of range", ids[p]);
        any_long |= ctx->lens[ids[p]] > 2048;
    }
    LaneGuard guard(ctx, LaneGuard::ANY);
    Lane& L = guard.lane();
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(L.d_out.reserve((size_t)count * elem_size));
    if (any_long) { // the long-ref kernel keeps its 2-D grid: list by list
        double ms = 0;
        int launches = 0;
        for (int32_t g = 0; g < n_groups; ++g) {
            const int32_t m = (int32_t)(group_offsets[g + 1] - group_offsets[g]);
            if (m < 2) continue;
            const int32_t* gi = ids + group_offsets[g];
            int rc = run_rows(ctx, L, lcsgpu::MODE_TRIANGLE, gi, 0, m, gi, 0, m - 1,
                              (char*)L.d_out.p + (size_t)tri_base[g] * elem_size, 0, 0, elem_size, 0);
            if (rc) return rc;
            HIP_TRY(hipStreamSynchronize(L.stream));
            finish_host_call(ctx, L);
            ms += g_last.ms;
            launches += g_last.launches;
        }
        HIP_TRY(hipMemcpy(out, L.d_out.p, (size_t)count * elem_size, hipMemcpyDeviceToHost));
        g_last.ms = ms;
        g_last.launches = launches;
        return LCSGPU_OK;
    }

    // per instantiated kernel: its refs (in position order, so the refs of one list are adjacent) and its jobs
    struct BatchBucket {
        int bv;
        bool quirk;
        std::vector<int32_t> ref_id, ref_col0;
        std::vector<int64_t> ref_row, ref_out0;
        std::vector<int32_t> ref_group;
        std::vector<int4> jobs;
        int refs_per_wg = 0;
    };
    std::vector<BatchBucket> buckets;
    int index_of[160];
    std::fill(index_of, index_of + 160, -1);
    for (int32_t g = 0; g < n_groups; ++g)
        for (int64_t p = group_offsets[g]; p < group_offsets[g + 1]; ++p) {
            if (p == group_offsets[g]) continue; // the first member of a list has no partner
            const int32_t id = ids[p];
            const bool q = ctx->quirk[id] != 0;
            const int bv = q ? lcsgpu::quirk_h_class(ctx->lens[id]) : lcsgpu::h_class(ctx->lens[id]);
            const int key = bv * 2 + (q ? 1 : 0);
            if (index_of[key] < 0) {
                index_of[key] = (int)buckets.size();
                buckets.push_back(BatchBucket{bv, q, {}, {}, {}, {}, {}, {}, 0});
            }
            BatchBucket& b = buckets[index_of[key]];
            b.ref_id.push_back(id);
            b.ref_row.push_back(p);
            b.ref_col0.push_back((int32_t)group_offsets[g]);
            b.ref_out0.push_back(tri_base[g]);
            b.ref_group.push_back(g);
        }
    size_t bytes = 0;
    auto align16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t col_off = 0;
    bytes += align16((size_t)n_total * 4);
    std::vector<size_t> o_id(buckets.size()), o_row(buckets.size()), o_c0(buckets.size()), o_out(buckets.size()),
        o_job(buckets.size());
    for (size_t bi = 0; bi < buckets.size(); ++bi) {
        BatchBucket& b = buckets[bi];
        const int R = b.refs_per_wg = lcsgpu::refs_per_block_for(b.bv, b.quirk, (long)b.ref_id.size(), 1);
        const size_t nr_all = b.ref_id.size();
        for (size_t k0 = 0; k0 < nr_all;) {
            size_t k1 = k0 + 1;
            while (k1 < nr_all && k1 - k0 < (size_t)R && b.ref_group[k1] == b.ref_group[k0]) ++k1;
            const int32_t g0 = b.ref_col0[k0];
            const int32_t max_row = (int32_t)b.ref_row[k1 - 1]; // rows ascend inside a list
            for (int32_t c0 = g0; c0 < max_row; c0 += 256)
                b.jobs.push_back(make_int4((int)k0, (int)(k1 - k0), c0, max_row));
            k0 = k1;
        }
        if (b.jobs.size() > 0x7fffffffu) return fail(LCSGPU_E_INVALID, "batch too large");
        o_id[bi] = bytes; bytes += align16(nr_all * 4);
        o_row[bi] = bytes; bytes += align16(nr_all * 8);
        o_c0[bi] = bytes; bytes += align16(nr_all * 4);
        o_out[bi] = bytes; bytes += align16(nr_all * 8);
        o_job[bi] = bytes; bytes += align16(b.jobs.size() * sizeof(int4));
    }
    if (L.plan_in_flight) {
        HIP_TRY(hipStreamSynchronize(L.stream));
        L.plan_in_flight = false;
    }
    HIP_TRY(L.h_plan.reserve(bytes));
    HIP_TRY(L.d_plan.reserve(bytes));
    char* h = (char*)L.h_plan.p;
    memcpy(h + col_off, ids, (size_t)n_total * 4);
    for (size_t bi = 0; bi < buckets.size(); ++bi) {
        const BatchBucket& b = buckets[bi];
        memcpy(h + o_id[bi], b.ref_id.data(), b.ref_id.size() * 4);
        memcpy(h + o_row[bi], b.ref_row.data(), b.ref_row.size() * 8);
        memcpy(h + o_c0[bi], b.ref_col0.data(), b.ref_col0.size() * 4);
        memcpy(h + o_out[bi], b.ref_out0.data(), b.ref_out0.size() * 8);
        memcpy(h + o_job[bi], b.jobs.data(), b.jobs.size() * sizeof(int4));
    }
    HIP_TRY(hipMemcpyAsync(L.d_plan.p, h, bytes, hipMemcpyHostToDevice, L.stream));
    L.plan_in_flight = true;
    L.last_launches = 0;
    HIP_TRY(hipEventRecord(L.ev_start, L.stream));
    for (size_t bi = 0; bi < buckets.size(); ++bi) {
        const BatchBucket& b = buckets[bi];
        if (b.jobs.empty()) continue;
        RowsArgs a{};
        a.tiles = (const uint8_t*)ctx->d_tiles.p;
        a.tile_base = (const uint64_t*)ctx->d_tile_base.p;
        a.lens = (const uint32_t*)ctx->d_lens.p;
        a.n_refs = (int32_t)b.ref_id.size();
        char* d = (char*)L.d_plan.p;
        a.ref_ids = (const int32_t*)(d + o_id[bi]);
        a.ref_rows = (const int64_t*)(d + o_row[bi]);
        a.ref_col0 = (const int32_t*)(d + o_c0[bi]);
        a.ref_out0 = (const int64_t*)(d + o_out[bi]);
        a.jobs = (const int4*)(d + o_job[bi]);
        a.col_ids = (const int32_t*)(d + col_off);
        a.n_cols = (int32_t)n_total;
        a.out = L.d_out.p;
        a.elem_size = elem_size;
        a.mode = lcsgpu::MODE_TRIANGLE;
        a.refs_per_block = b.refs_per_wg;
        HIP_TRY(lcsgpu::launch_rows(b.bv, b.quirk, a, (int)b.jobs.size(), 1, L.stream));
        ++L.last_launches;
    }
    HIP_TRY(hipEventRecord(L.ev_stop, L.stream));
    L.timing_valid = true;
    HIP_TRY(hipMemcpyAsync(out, L.d_out.p, (size_t)count * elem_size, hipMemcpyDeviceToHost, L.stream));
    HIP_TRY(hipEventRecord(L.ev_done, L.stream));
    HIP_TRY(hipEventSynchronize(L.ev_done));
    finish_host_call(ctx, L);
    return LCSGPU_OK;
}

int lcsgpu_assign_seeds(lcsgpu_ctx* ctx, const int32_t* seed_ids, int32_t n_seeds, const int32_t* col_ids,
                        int32_t n_cols, int distance_kind, int32_t first_k, float* dist, int32_t* assign)
{
    if (!ctx) return fail(LCSGPU_E_INVALID, "NULL ctx");
    if (ctx->n < 0) return fail(LCSGPU_E_STATE, "no sequence set uploaded");
    if (distance_kind != LCSGPU_DIST_INDEL_DIV_LCS && distance_kind != LCSGPU_DIST_INDEL075_DIV_LCS)
        return fail(LCSGPU_E_INVALID, "unknown distance kind %d", distance_kind);
    if (n_seeds < 0 || n_cols < 0) return fail(LCSGPU_E_INVALID, "negative count");
    if (n_seeds == 0 || n_cols == 0) return LCSGPU_OK;
    if (!seed_ids || !col_ids || !dist || !assign) return fail(LCSGPU_E_INVALID, "NULL argument");
    for (int32_t r = 0; r < n_seeds; ++r)
        if (seed_ids[r] < 0 || seed_ids[r] >= ctx->n) return fail(LCSGPU_E_INVALID, "seed id %d out of range", seed_ids[r]);
    LaneGuard guard(ctx, LaneGuard::ANY);
    Lane& L = guard.lane();
    HIP_TRY(hipSetDevice(ctx->device));
    const int elem = ctx->max_len > 65535 ? 4 : 2;
    // column chunks: the LCS rectangle of a chunk stays below 256 MB
    const int32_t chunk = (int32_t)std::max<int64_t>(4096, std::min<int64_t>(n_cols, ((int64_t)256 << 20) / ((int64_t)n_seeds * elem)));
    auto a256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_seeds = 0, o_cols = o_seeds + a256((size_t)n_seeds * 4), o_dist = o_cols + a256((size_t)chunk * 4),
                 o_assign = o_dist + a256((size_t)chunk * 4), total = o_assign + a256((size_t)chunk * 4);
    HIP_TRY(L.d_work.reserve(total));
    HIP_TRY(L.d_out.reserve((size_t)n_seeds * chunk * elem));
    char* base = (char*)L.d_work.p;
    HIP_TRY(hipMemcpyAsync(base + o_seeds, seed_ids, (size_t)n_seeds * 4, hipMemcpyHostToDevice, L.stream));
    double ms = 0;
    int launches = 0;
    for (int32_t c0 = 0; c0 < n_cols; c0 += chunk) {
        const int32_t cn = std::min(chunk, n_cols - c0);
        HIP_TRY(hipMemcpyAsync(base + o_cols, col_ids + c0, (size_t)cn * 4, hipMemcpyHostToDevice, L.stream));
        HIP_TRY(hipMemcpyAsync(base + o_dist, dist + c0, (size_t)cn * 4, hipMemcpyHostToDevice, L.stream));
        HIP_TRY(hipMemcpyAsync(base + o_assign, assign + c0, (size_t)cn * 4, hipMemcpyHostToDevice, L.stream));
        int rc = run_rows(ctx, L, lcsgpu::MODE_RECT, seed_ids, 0, n_seeds, col_ids + c0, 0, cn, L.d_out.p, cn, 0, elem);
        if (rc) return rc;
        HIP_TRY(lcsgpu::launch_assign_seeds(L.d_out.p, elem, cn, (const int32_t*)(base + o_seeds), n_seeds,
                                            (const int32_t*)(base + o_cols), cn, (const uint32_t*)ctx->d_lens.p,
                                            (const float*)ctx->d_powf.p, distance_kind, first_k, (float*)(base + o_dist),
                                            (int32_t*)(base + o_assign), L.stream));
        HIP_TRY(hipMemcpyAsync(dist + c0, base + o_dist, (size_t)cn * 4, hipMemcpyDeviceToHost, L.stream));
        HIP_TRY(hipMemcpyAsync(assign + c0, base + o_assign, (size_t)cn * 4, hipMemcpyDeviceToHost, L.stream));
        HIP_TRY(hipEventRecord(L.ev_done, L.stream));
        HIP_TRY(hipEventSynchronize(L.ev_done));
        finish_host_call(ctx, L);
        ms += g_last.ms;
        launches += g_last.launches;
    }
    g_last.ms = ms;
    g_last.launches = launches;
    return LCSGPU_OK;
}

int lcsgpu_clarans(lcsgpu_ctx* ctx, const int32_t* ids, int32_t n_ids, int distance_kind, int32_t n_medoids,
                   int32_t n_fixed, float explore_fraction, int32_t num_local, int32_t* medoids_out)
{
    if (!ctx) return fail(LCSGPU_E_INVALID, "NULL ctx");
    if (ctx->n < 0) return fail(LCSGPU_E_STATE, "no sequence set uploaded");
    if (distance_kind != LCSGPU_DIST_INDEL_DIV_LCS && distance_kind != LCSGPU_DIST_INDEL075_DIV_LCS)
        return fail(LCSGPU_E_INVALID, "unknown distance kind %d", distance_kind);
    if (!ids || !medoids_out || n_ids < 1) return fail(LCSGPU_E_INVALID, "bad sample / output");
    if (n_medoids < 1 || n_medoids > n_ids || n_fixed < 0 || n_fixed >= n_medoids || num_local < 1)
        return fail(LCSGPU_E_INVALID, "bad CLARANS shape: %d medoids (%d fixed) of %d, %d searches", n_medoids, n_fixed,
                    n_ids, num_local);
    for (int32_t i = 0; i < n_ids; ++i)
        if (ids[i] < 0 || ids[i] >= ctx->n) return fail(LCSGPU_E_INVALID, "sample id %d out of range", ids[i]);
    if (n_medoids > lcsgpu::CLARANS_MAX_MEDOIDS)
        return fail(LCSGPU_E_UNSUPPORTED, "device CLARANS handles at most %d medoids", lcsgpu::CLARANS_MAX_MEDOIDS);

    const int32_t n = n_ids, k = n_medoids;
    // Clustering.cpp:21-29: how many non-improving steps end a local search
    const int n_swaps = (n - k) * k;
    const int min_max_neighbor = 250;
    const int max_neighbor = n_swaps < min_max_neighbor
                                 ? n_swaps
                                 : std::max((int)(explore_fraction * n_swaps), min_max_neighbor);
    const int corrected = max_neighbor / k;

    LaneGuard guard(ctx, LaneGuard::ANY);
    Lane& L = guard.lane();
    HIP_TRY(hipSetDevice(ctx->device));
    const int elem = ctx->max_len > 65535 ? 4 : 2;
    const size_t pairs = (size_t)n * (n - 1) / 2;
    auto a256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t window = (size_t)std::max(corrected, 1);
    const size_t o_D = 0, o_DM = o_D + a256(pairs * 4), o_cand = o_DM + a256((size_t)n * k * 4),
                 o_st = o_cand + a256((size_t)n * 4), o_rd = o_st + a256((size_t)n * 16), o_rm = o_rd + a256(std::max<size_t>(window, 64) * 4),
                 o_wxx = o_rm + a256(std::max<size_t>(window, 64) * 4), o_wx = o_wxx + a256(window * 8), o_log = o_wx + a256(window * 8), o_state = o_log + a256((size_t)(n + 1) * 4), o_ids = o_state + 256,
                 total = o_ids + a256((size_t)n * 4);
    HIP_TRY(L.d_work.reserve(total));
    HIP_TRY(L.h_small.reserve(64));
    char* base = (char*)L.d_work.p;
    HIP_TRY(hipMemsetAsync(base + o_state, 0, 256, L.stream));
    HIP_TRY(hipMemcpyAsync(base + o_ids, ids, (size_t)n * 4, hipMemcpyHostToDevice, L.stream));
    if (pairs > 0) {
        HIP_TRY(L.d_out.reserve(pairs * elem));
        int rc = run_rows(ctx, L, lcsgpu::MODE_TRIANGLE, ids, 0, n, ids, 0, n - 1, L.d_out.p, 0, 0, elem, 0);
        if (rc) return rc;
        HIP_TRY(lcsgpu::launch_subset_distances(L.d_out.p, elem, (const int32_t*)(base + o_ids),
                                                (const uint32_t*)ctx->d_lens.p, (const float*)ctx->d_powf.p,
                                                distance_kind, n, (float*)(base + o_D), L.stream));
    }
    lcsgpu::ClaransArgs a{};
    a.D = (const float*)(base + o_D);
    a.DMt = (float*)(base + o_DM);
    a.cand = (int32_t*)(base + o_cand);
    a.st = (float4*)(base + o_st);
    a.res_delta = (float*)(base + o_rd);
    a.res_mm = (int32_t*)(base + o_rm);
    a.win_xx = (int32_t*)(base + o_wxx);
    a.win_x = (int32_t*)(base + o_wx);
    a.win_cap = (int32_t)window;
    a.cost_log = (float*)(base + o_log);
    a.state = (int32_t*)(base + o_state);
    a.n_elems = n;
    a.n_medoids = k;
    a.n_fixed = n_fixed;

    a.corrected = corrected;
    // The two generators of Clustering.cpp:43-44.  Neither looks at the search state, so the host
    // runs them: gen_nodes shuffles the candidate order before every local search, gen_positions
    // yields the step positions, handed to the device as a growing array of draws.
    std::mt19937 gen_nodes, gen_positions;
    std::vector<int32_t> cand(n), draws;
    for (int32_t i = 0; i < n; ++i) cand[i] = i;
    ClaransJob job;
    job.a = a;
    job.gen_positions = &gen_positions;
    job.draws = &draws;
    job.d_draws = &L.d_draws;
    float best_cost = std::numeric_limits<float>::max();
    for (int iter = 0; iter < num_local; ++iter) {
        // partial_shuffle(candidate + n_fixed, candidate + n, candidate + n, gen_nodes), deterministic_random.h:113-127
        {
            int32_t* first = cand.data() + n_fixed;
            const long cnt = n - n_fixed, N = cnt - 1;
            for (long i = 0; i < cnt; ++i) {
                const unsigned long d = (unsigned long)N - (unsigned long)i + 1;
                const unsigned long r = (unsigned long)gen_nodes(); // < 2^32: never in the rejected tail of a 64-bit range
                std::swap(first[i], first[(r % d) + (unsigned long)i]);
            }
        }
        HIP_TRY(hipMemcpyAsync(a.cand, cand.data(), (size_t)n * 4, hipMemcpyHostToDevice, L.stream));
        if (n > k) { // the init kernel already needs the first window's draws
            int rc = clarans_extend_draws(job, (size_t)job.p_host + (size_t)std::max(corrected, 1), L.stream);
            if (rc) return rc;
        }
        HIP_TRY(lcsgpu::launch_clarans_init(job.a, L.stream));
        HIP_TRY(hipEventRecord(L.ev_done, L.stream));
        HIP_TRY(hipEventSynchronize(L.ev_done)); // the rounds run on the batch stream
        L.plan_in_flight = false;
        int rc = clarans_run_search(ctx, job);
        if (rc) return rc;
        float cost;
        memcpy(&cost, &job.state[5], 4);
        HIP_TRY(hipMemcpy(cand.data(), a.cand, (size_t)n * 4, hipMemcpyDeviceToHost));
        if (cost < best_cost) {
            best_cost = cost;
            std::copy(cand.begin(), cand.begin() + k, medoids_out);
        }
    }
    finish_host_call(ctx, L);
    return LCSGPU_OK;
}

int lcsgpu_sync(lcsgpu_ctx* ctx)
{
    if (!ctx) return fail(LCSGPU_E_INVALID, "NULL ctx");
    LaneGuard guard(ctx, LaneGuard::LANE0);
    Lane& L = guard.lane();
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipStreamSynchronize(L.stream));
    L.plan_in_flight = false;
    return LCSGPU_OK;
}

int lcsgpu_last_kernel_ms(lcsgpu_ctx* ctx, double* ms, int32_t* n_launches)
{
    if (!ctx || !ms) return fail(LCSGPU_E_INVALID, "NULL argument");
    *ms = 0.0;
    if (n_launches) *n_launches = 0;
    if (g_last.ctx != ctx) return LCSGPU_OK;
    if (!g_last.pending_on_lane0) { // a completed host-memory call of this thread
        *ms = g_last.ms;
        if (n_launches) *n_launches = g_last.launches;
        return LCSGPU_OK;
    }
    LaneGuard guard(ctx, LaneGuard::LANE0);
    Lane& L = guard.lane();
    if (n_launches) *n_launches = L.last_launches;
    if (!L.timing_valid || L.last_launches == 0) return LCSGPU_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventSynchronize(L.ev_stop));
    float f = 0.f;
    HIP_TRY(hipEventElapsedTime(&f, L.ev_start, L.ev_stop));
    *ms = (double)f;
    return LCSGPU_OK;
}

int lcsgpu_total_kernel_ms(lcsgpu_ctx* ctx, double* ms)
{
    if (!ctx || !ms) return fail(LCSGPU_E_INVALID, "NULL argument");
    std::lock_guard<std::mutex> lk(ctx->mu);
    *ms = ctx->total_kernel_ms;
    return LCSGPU_OK;
}

void* lcsgpu_stream(lcsgpu_ctx* ctx) { return ctx && !ctx->lanes.empty() ? (void*)ctx->lanes[0].stream : nullptr; }

} // extern "C"
