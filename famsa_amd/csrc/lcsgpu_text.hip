// lcsgpu_text.hip -- -dist_export as text: the C-ABI calls lcsgpu_dist_text_* (include/lcsgpu.h).
//
// A row block goes through the device as: LCS rectangle (columns in length order, so that a wave of the LCS kernels
// streams partners of one length) -> the block's final bytes (text_kernels.hip) -> pinned host memory.  A context holds
// a few SLOTS so that the caller can keep several blocks in flight: all slots queue their kernels on the context's
// stream (lane 0's: the blocks are computed one after the other, the chip is idle most of the stage anyway), a block's
// text leaves on the copy stream while the next block is computed, and the caller writes a finished block to its file
// while both go on.  Nothing here touches a file.
#include "lcsgpu_internal.h"

using namespace lcsgpu_impl;

struct TextSlot {
    Lane lane;            // what run_rows needs of a lane: staging for its plan, events; the stream is lane 0's
    DevBuf d_lcs, d_text, d_where, d_scan;
    PinBuf h_text, h_small; // h_small: where[] going up (n x 4 B), the block's byte count coming back (8 B at the front)
    hipEvent_t ev_ready = nullptr, ev_copied = nullptr;
    std::vector<int32_t> refs, cols;
    int32_t r0 = 0, r1 = 0;
    bool pending = false;
};
struct TextExport {
    int kind = 1, square = 0;
    std::vector<TextSlot> slots;
    DevBuf d_ids, d_id_off;
    std::vector<uint64_t> id_off;
    std::vector<int32_t> by_length; // the set's sequences, longest first (stable)
    hipStream_t copy_stream = nullptr;
};

namespace lcsgpu_impl {

void text_release(lcsgpu_ctx* ctx)
{
    TextExport* x = ctx->text;
    if (!x) return;
    ctx->text = nullptr;
    (void)hipSetDevice(ctx->device);
    if (x->copy_stream) (void)hipStreamSynchronize(x->copy_stream);
    if (ctx->lanes[0].stream) (void)hipStreamSynchronize(ctx->lanes[0].stream);
    for (TextSlot& s : x->slots) {
        s.d_lcs.release();
        s.d_text.release();
        s.d_where.release();
        s.d_scan.release();
        s.h_text.release();
        s.h_small.release();
        s.lane.d_plan.release();
        s.lane.d_carry.release();
        s.lane.h_plan.release();
        if (s.lane.ev_start) (void)hipEventDestroy(s.lane.ev_start);
        if (s.lane.ev_stop) (void)hipEventDestroy(s.lane.ev_stop);
        if (s.ev_ready) (void)hipEventDestroy(s.ev_ready);
        if (s.ev_copied) (void)hipEventDestroy(s.ev_copied);
    }
    x->d_ids.release();
    x->d_id_off.release();
    if (x->copy_stream) (void)hipStreamDestroy(x->copy_stream);
    delete x;
}

} // namespace lcsgpu_impl

extern "C" {

int lcsgpu_dist_text_begin(lcsgpu_ctx* ctx, const char* ids, const uint64_t* id_offsets, int distance_kind, int flags,
                           int32_t n_slots)
{
    if (!ctx || !id_offsets) return fail(LCSGPU_E_INVALID, "NULL argument");
    if (ctx->n < 0) return fail(LCSGPU_E_STATE, "no sequence set uploaded");
    const bool pid = (flags & LCSGPU_TEXT_PID) != 0;
    if (!pid && distance_kind != LCSGPU_DIST_INDEL_DIV_LCS && distance_kind != LCSGPU_DIST_INDEL075_DIV_LCS)
        return fail(LCSGPU_E_INVALID, "unknown distance kind %d", distance_kind);
    if (n_slots < 1 || n_slots > 8) return fail(LCSGPU_E_INVALID, "1 .. 8 slots");
    const int32_t n = ctx->n;
    for (int32_t i = 0; i < n; ++i)
        if (id_offsets[i + 1] < id_offsets[i]) return fail(LCSGPU_E_INVALID, "id offsets not monotone at %d", i);
    if (n > 0 && id_offsets[n] > id_offsets[0] && !ids) return fail(LCSGPU_E_INVALID, "NULL ids");
    LaneGuard guard(ctx, LaneGuard::LANE0);
    HIP_TRY(hipSetDevice(ctx->device));
    text_release(ctx);
    TextExport* x = new (std::nothrow) TextExport;
    if (!x) return fail(LCSGPU_E_NOMEM, "out of host memory");
    ctx->text = x;
    x->kind = pid ? 2 : distance_kind;
    x->square = (flags & LCSGPU_TEXT_SQUARE) ? 1 : 0;
    x->slots.resize((size_t)n_slots);
    x->id_off.assign(id_offsets, id_offsets + n + 1);
    const uint64_t base = n > 0 ? id_offsets[0] : 0, id_bytes = n > 0 ? id_offsets[n] - base : 0;
    for (uint64_t& o : x->id_off) o -= base;
    hipStream_t st = ctx->lanes[0].stream;
    HIP_TRY(x->d_ids.reserve((size_t)id_bytes + 16));
    HIP_TRY(x->d_id_off.reserve(((size_t)n + 1) * 8));
    if (id_bytes) HIP_TRY(hipMemcpyAsync(x->d_ids.p, ids + base, (size_t)id_bytes, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemcpyAsync(x->d_id_off.p, x->id_off.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice, st));
    x->by_length.resize((size_t)n);
    for (int32_t i = 0; i < n; ++i) x->by_length[i] = i;
    std::stable_sort(x->by_length.begin(), x->by_length.end(), [&](int32_t a, int32_t b) { return ctx->lens[a] > ctx->lens[b]; });
    HIP_TRY(hipStreamCreateWithFlags(&x->copy_stream, hipStreamNonBlocking));
    for (TextSlot& s : x->slots) {
        s.lane.stream = st;
        HIP_TRY(hipEventCreate(&s.lane.ev_start));
        HIP_TRY(hipEventCreate(&s.lane.ev_stop));
        HIP_TRY(hipEventCreateWithFlags(&s.ev_ready, hipEventBlockingSync | hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&s.ev_copied, hipEventBlockingSync | hipEventDisableTiming));
    }
    HIP_TRY(hipStreamSynchronize(st)); // x->id_off may now be read by nobody but the host
    return LCSGPU_OK;
}

int lcsgpu_dist_text_submit(lcsgpu_ctx* ctx, int32_t slot, int32_t row_begin, int32_t row_end)
{
    if (!ctx) return fail(LCSGPU_E_INVALID, "NULL ctx");
    TextExport* x = ctx->text;
    if (!x) return fail(LCSGPU_E_STATE, "lcsgpu_dist_text_begin first");
    if (slot < 0 || slot >= (int32_t)x->slots.size()) return fail(LCSGPU_E_INVALID, "no slot %d", slot);
    const int32_t n = ctx->n;
    if (row_begin < 0 || row_end <= row_begin || row_end > n) return fail(LCSGPU_E_INVALID, "bad row range");
    if (row_end - row_begin > 32768) return fail(LCSGPU_E_INVALID, "at most 32768 rows per block");
    TextSlot& s = x->slots[(size_t)slot];
    if (s.pending) return fail(LCSGPU_E_STATE, "slot %d holds a block that has not been waited for", slot);
    const int32_t n_rows = row_end - row_begin;
    const int elem = ctx->max_len > 65535 ? 4 : 2;
    // the rectangle's columns: every sequence a row of the block needs, longest first
    const int32_t col_limit = x->square ? n : row_end - 1;
    s.cols.clear();
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(s.h_small.reserve(16 + (size_t)n * 4));
    int32_t* where = (int32_t*)((char*)s.h_small.p + 16);
    for (int32_t j : x->by_length)
        if (j < col_limit) {
            where[j] = (int32_t)s.cols.size();
            s.cols.push_back(j);
        }
    const int32_t n_cols = (int32_t)s.cols.size();
    s.refs.resize((size_t)n_rows);
    for (int32_t r = 0; r < n_rows; ++r) s.refs[(size_t)r] = row_begin + r;
    const int seg_values = lcsgpu::text_segment_values();
    const int32_t max_cols = x->square ? n : row_end - 1;
    const int32_t segs = std::max(1, (max_cols + seg_values - 1) / seg_values);
    // device text: sized for the longest possible values, so that no block can overflow it
    uint64_t worst = 0;
    for (int32_t i = row_begin; i < row_end; ++i)
        worst += (x->id_off[i + 1] - x->id_off[i]) + 1 + (uint64_t)(x->square ? n : i) * (uint64_t)lcsgpu::text_max_value_bytes();
    int rc = reserve_big(ctx, s.d_text, (size_t)worst + 64, "dist_text block");
    if (rc) return rc;
    if (n_cols > 0) {
        rc = reserve_big(ctx, s.d_lcs, (size_t)n_rows * (size_t)n_cols * elem, "dist_text rectangle");
        if (rc) return rc;
    }
    const size_t scan_seg = ((size_t)n_rows * segs * 4 + 15) & ~(size_t)15, scan_len = ((size_t)n_rows * 4 + 15) & ~(size_t)15;
    HIP_TRY(s.d_scan.reserve(scan_seg + scan_len + ((size_t)n_rows + 1) * 8));
    HIP_TRY(s.d_where.reserve((size_t)n * 4 + 16));

    LaneGuard guard(ctx, LaneGuard::LANE0); // the context's stream, in call order with everything else queued there
    hipStream_t st = ctx->lanes[0].stream;
    s.lane.stream = st;
    if (n_cols > 0) {
        HIP_TRY(hipMemcpyAsync(s.d_where.p, where, (size_t)n * 4, hipMemcpyHostToDevice, st));
        rc = run_rows(ctx, s.lane, lcsgpu::MODE_RECT, s.refs.data(), 0, n_rows, s.cols.data(), 0, n_cols, s.d_lcs.p, n_cols, 0, elem);
        if (rc) return rc;
    } else {
        s.lane.last_launches = 0;
        s.lane.timing_valid = false;
    }
    lcsgpu::TextArgs t{};
    t.lcs = s.d_lcs.p;
    t.ld = n_cols;
    t.where = (const int32_t*)s.d_where.p;
    t.lens = (const uint32_t*)ctx->d_lens.p;
    t.pow_f64 = (const double*)ctx->d_pow.p;
    t.ids = (const char*)x->d_ids.p;
    t.id_off = (const uint64_t*)x->d_id_off.p;
    t.seg_len = (uint32_t*)s.d_scan.p;
    t.row_len = (uint32_t*)((char*)s.d_scan.p + scan_seg);
    t.row_start = (unsigned long long*)((char*)s.d_scan.p + scan_seg + scan_len);
    t.out = (char*)s.d_text.p;
    t.elem_size = elem;
    t.kind = x->kind;
    t.row_begin = row_begin;
    t.n_rows = n_rows;
    t.n = n;
    t.square = x->square;
    t.segs = segs;
    HIP_TRY(lcsgpu::launch_text_block(t, st));
    HIP_TRY(hipMemcpyAsync(s.h_small.p, t.row_start + n_rows, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipEventRecord(s.ev_ready, st));
    s.r0 = row_begin;
    s.r1 = row_end;
    s.pending = true;
    return LCSGPU_OK;
}

int lcsgpu_dist_text_wait(lcsgpu_ctx* ctx, int32_t slot, const char** text, uint64_t* n_bytes)
{
    if (!ctx || !text || !n_bytes) return fail(LCSGPU_E_INVALID, "NULL argument");
    *text = nullptr;
    *n_bytes = 0;
    TextExport* x = ctx->text;
    if (!x) return fail(LCSGPU_E_STATE, "lcsgpu_dist_text_begin first");
    if (slot < 0 || slot >= (int32_t)x->slots.size()) return fail(LCSGPU_E_INVALID, "no slot %d", slot);
    TextSlot& s = x->slots[(size_t)slot];
    if (!s.pending) return fail(LCSGPU_E_STATE, "nothing submitted on slot %d", slot);
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventSynchronize(s.ev_ready)); // the block's text is in HBM, its size here
    s.pending = false;
    s.lane.plan_in_flight = false;
    const uint64_t bytes = *(const volatile uint64_t*)s.h_small.p;
    if (bytes + 64 > s.d_text.cap) return fail(LCSGPU_E_HIP, "dist_text: a block of %llu bytes in a buffer of %zu", (unsigned long long)bytes, s.d_text.cap);
    float ms = 0.f;
    if (s.lane.timing_valid && s.lane.last_launches > 0 && hipEventElapsedTime(&ms, s.lane.ev_start, s.lane.ev_stop) == hipSuccess) {
        std::lock_guard<std::mutex> lk(ctx->mu);
        ctx->total_kernel_ms += ms;
    }
    note_host_call(ctx, ms, s.lane.last_launches); // lcsgpu_last_kernel_ms: the LCS launches of this block
    HIP_TRY(s.h_text.reserve((size_t)bytes + 16));
    if (bytes) {
        HIP_TRY(hipMemcpyAsync(s.h_text.p, s.d_text.p, (size_t)bytes, hipMemcpyDeviceToHost, x->copy_stream));
        HIP_TRY(hipEventRecord(s.ev_copied, x->copy_stream));
        HIP_TRY(hipEventSynchronize(s.ev_copied));
    }
    *text = (const char*)s.h_text.p;
    *n_bytes = bytes;
    return LCSGPU_OK;
}

int lcsgpu_dist_text_end(lcsgpu_ctx* ctx)
{
    if (!ctx) return fail(LCSGPU_E_INVALID, "NULL ctx");
    LaneGuard guard(ctx, LaneGuard::LANE0);
    text_release(ctx);
    return LCSGPU_OK;
}

} // extern "C"
