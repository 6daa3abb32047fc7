// lcsgpu_api.hip -- the C-ABI of include/lcsgpu.h over the gfx950 kernels.
// Context, HBM layout of the uploaded sequence set, launch planning.  No CPU compute path:
// every LCS value this library returns was produced by a HIP kernel.
#include "lcsgpu_internal.h"

namespace lcsgpu_impl {

thread_local std::string g_err;
thread_local LastCall g_last;

int tune_int(const char* key, int dflt)
{
    const char* e = getenv("LCSGPU_TUNE");
    if (!e) return dflt;
    const size_t kl = strlen(key);
    for (const char* p = e; *p;) {
        const char* end = strchr(p, ',');
        const size_t len = end ? (size_t)(end - p) : strlen(p);
        if (len > kl + 1 && !strncmp(p, key, kl) && p[kl] == '=') return atoi(p + kl + 1);
        p += len + (end ? 1 : 0);
    }
    return dflt;
}


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

// Which instantiation the refs of every half-word class run in.  A call's classes are launches one after the other on
// one stream, and a launch that cannot fill the chip lasts as long as one of its workgroups however few it has: the
// sample triangle of a FastTree split (2000 family members, four classes of ~500 refs, ~500 workgroups each) took four
// times ~160 us (profiles/clarans_rounds_r04.txt: 4373 such launches, 1.22 s of kernel time for work that fills the chip
// for 0.3 s).  So neighbouring classes are run together in the larger one's kernel -- a ref's half-words beyond its own
// are all-ones no-ops (h_class) -- as long as the group stays below the workgroup count a launch is planned for
// (refs_per_block_for: 2048).  wgs[h] = workgroups class h would have with the fewest refs per workgroup; classes that
// fill the chip on their own (every large triangle) are left alone, as are the orientation-sensitive and long refs.
void merge_small_classes(const double* wgs, int* target)
{
    const double want = 2048.0;
    for (int h = 0; h <= 64; ++h) target[h] = h;
    int first = -1; // first class of the open group
    double sum = 0;
    auto close = [&](int last) {
        if (first > 0)
            for (int h = first; h <= last; ++h)
                if (wgs[h] > 0) target[h] = last;
        first = -1;
        sum = 0;
    };
    int last_used = -1;
    for (int h = 1; h <= 64; ++h) {
        if (wgs[h] <= 0) continue;
        if (first > 0 && sum + wgs[h] > want) close(last_used);
        if (wgs[h] >= want) { // fills the chip by itself
            close(last_used);
            last_used = h;
            continue;
        }
        if (first < 0) first = h;
        sum += wgs[h];
        last_used = h;
    }
    close(last_used);
}

// Split the refs by instantiated kernel (word-count class, quirk flag), keeping order.  col_blocks > 0: the
// column blocks a ref meets, for merge_small_classes; 0 = every class its own launch.
int make_buckets(lcsgpu_ctx* ctx, const int32_t* ref_ids, int32_t ref_begin, int32_t n_refs,
                 int64_t row0, std::vector<Bucket>& out, long col_blocks = 0)
{
    int index_of[160];
    std::fill(index_of, index_of + 160, -1);
    int target[65];
    for (int h = 0; h <= 64; ++h) target[h] = h;
    if (col_blocks > 0) {
        double wgs[65] = {0};
        for (int32_t k = 0; k < n_refs; ++k) {
            const int32_t id = ref_ids ? ref_ids[k] : ref_begin + k;
            if (id < 0 || id >= ctx->n)
                return fail(LCSGPU_E_INVALID, "ref id %d out of range [0,%d)", id, ctx->n);
            if (ctx->quirk[id]) continue;
            const int h = lcsgpu::h_class(ctx->lens[id]);
            if (h > 0) wgs[h] += (double)col_blocks / lcsgpu::refs_per_block_for(h, false, 1, 1);
        }
        merge_small_classes(wgs, target);
    }
    for (int32_t k = 0; k < n_refs; ++k) {
        const int32_t id = ref_ids ? ref_ids[k] : ref_begin + k;
        if (id < 0 || id >= ctx->n)
            return fail(LCSGPU_E_INVALID, "ref id %d out of range [0,%d)", id, ctx->n);
        const bool q = ctx->quirk[id] != 0;
        // bv = instantiated half-word count; 0 = the long-sequence kernel (> 2048 residues)
        const int bv = q ? lcsgpu::quirk_h_class(ctx->lens[id]) : target[lcsgpu::h_class(ctx->lens[id])];
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

bool create_lane(lcsgpu_ctx* ctx, Lane& l, bool high_priority)
{
    if (hipSetDevice(ctx->device) != hipSuccess) return false;
    if (high_priority) // a lane that works while other lanes keep the GPU fed: no hipFree when a buffer grows
        l.d_plan.keep_outgrown = l.d_out.keep_outgrown = l.d_carry.keep_outgrown = l.d_work.keep_outgrown = l.d_draws.keep_outgrown =
            l.h_plan.keep_outgrown = l.h_small.keep_outgrown = true;
    if (high_priority) {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || least == greatest) {
            (void)hipGetLastError();
            return false;
        }
        const bool ok = hipStreamCreateWithPriority(&l.stream, hipStreamNonBlocking, greatest) == hipSuccess &&
                        hipEventCreate(&l.ev_start) == hipSuccess && hipEventCreate(&l.ev_stop) == hipSuccess &&
                        hipEventCreateWithFlags(&l.ev_done, hipEventBlockingSync | hipEventDisableTiming) == hipSuccess;
        if (!ok) (void)hipGetLastError();
        return ok;
    }
    // (the copy stream of the sliced host-buffer triangle is made by its first user: a stream costs ~100 MB of resident
    //  host memory and some milliseconds on this runtime, and the FastTree recursion's 16+ lanes never need it)
    const bool ok = hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking) == hipSuccess &&
                    hipEventCreate(&l.ev_start) == hipSuccess && hipEventCreate(&l.ev_stop) == hipSuccess &&
                    hipEventCreateWithFlags(&l.ev_done, hipEventBlockingSync | hipEventDisableTiming) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    return ok;
}


// free device memory as reserve_big sees it (LCSGPU_FAKE_HBM_GB included)
int device_free_bytes(lcsgpu_ctx* ctx, size_t* out)
{
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(LCSGPU_E_HIP, "hipSetDevice(%d) failed", ctx->device);
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    if (const char* e = getenv("LCSGPU_FAKE_HBM_GB")) free_b = std::min(free_b, (size_t)(atof(e) * 1e9));
    *out = free_b;
    return LCSGPU_OK;
}

int reserve_big(lcsgpu_ctx* ctx, DevBuf& buf, size_t bytes, const char* what)
{
    if (bytes <= buf.cap) return LCSGPU_OK;
    if (hipSetDevice(ctx->device) != hipSuccess) return fail(LCSGPU_E_HIP, "hipSetDevice(%d) failed", ctx->device);
    size_t free_b = 0, total_b = 0;
    const bool known = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
    if (const char* e = getenv("LCSGPU_FAKE_HBM_GB")) // tests: behave as if only this much device memory were free
        free_b = std::min(free_b, (size_t)(atof(e) * 1e9));
    if (known && bytes > free_b + buf.cap)
        return fail(LCSGPU_E_NOMEM, "%s needs %.1f GB of device memory, %.1f GB are free on device %d (of %.1f GB)", what,
                    bytes / 1e9, (free_b + buf.cap) / 1e9, ctx->device, total_b / 1e9);
    const auto t0 = std::chrono::steady_clock::now();
    const hipError_t e = buf.reserve(bytes);
    if (bytes >= ((size_t)1 << 30) && getenv("LCSGPU_PROFILE")) // (right after another process has ended, 44 GB can take a second or two)
        fprintf(stderr, "lcsgpu: %.1f GB for %s allocated in %.3f s\n", bytes / 1e9, what,
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    if (e != hipSuccess) {
        (void)hipGetLastError(); // an allocation failure is not sticky
        return fail(LCSGPU_E_NOMEM, "%s: allocating %.1f GB of device memory failed: %s", what, bytes / 1e9,
                    hipGetErrorString(e));
    }
    return LCSGPU_OK;
}

// Core: plan + launch.  d_out is a device pointer.
int run_rows(lcsgpu_ctx* ctx, Lane& L, int mode, const int32_t* ref_ids, int32_t ref_begin, int32_t n_refs,
             const int32_t* col_ids, int32_t col_begin, int32_t n_cols, void* d_out, int64_t ld,
             int64_t out_offset, int elem_size, int64_t first_row, const lcsgpu::FuseArgs* fuse)
{
    if (ctx->n < 0) return fail(LCSGPU_E_STATE, "no sequence set uploaded");
    if (elem_size != 2 && elem_size != 4) return fail(LCSGPU_E_INVALID, "elem_size must be 2 or 4");
    if (fuse) {
        if (mode != lcsgpu::MODE_TRIANGLE || ref_ids || col_ids || first_row != ref_begin)
            return fail(LCSGPU_E_INVALID, "a fused launch covers contiguous rows of the triangle");
        if (ctx->max_len > 65535) return fail(LCSGPU_E_UNSUPPORTED, "fused records hold 16-bit lengths");
    }
    if (elem_size == 2 && ctx->max_len > 65535)
        return fail(LCSGPU_E_INVALID, "uint16 output needs all sequences <= 65535 residues");
    if (n_refs < 0 || n_cols < 0) return fail(LCSGPU_E_INVALID, "negative count");
    L.last_launches = 0;
    L.timing_valid = false;
    if (n_refs == 0 || n_cols == 0) return LCSGPU_OK;
    if (!fuse && !d_out) return fail(LCSGPU_E_INVALID, "NULL output");
    if (!col_ids && (col_begin < 0 || (int64_t)col_begin + n_cols > ctx->n))
        return fail(LCSGPU_E_INVALID, "column range out of bounds");
    if (col_ids)
        for (int32_t c = 0; c < n_cols; ++c)
            if (col_ids[c] < 0 || col_ids[c] >= ctx->n)
                return fail(LCSGPU_E_INVALID, "col id %d out of range", col_ids[c]);

    std::vector<Bucket> buckets;
    // (triangle: about half of the column blocks of a ref tile lie below the diagonal; a fused launch keeps its classes)
    const long call_col_blocks = std::max<long>(1, ((long)n_cols + 255) / 256 / (mode == lcsgpu::MODE_TRIANGLE ? 2 : 1));
    int rc = make_buckets(ctx, ref_ids, ref_begin, n_refs, first_row, buckets, fuse ? 0 : call_col_blocks);
    if (rc) return rc;

    // staging: [col_ids][per non-contiguous bucket: rows(int64) then ids(int32)]
    size_t bytes = 0;
    auto align8 = [](size_t x) { return (x + 7) & ~(size_t)7; };
    const size_t col_off = 0;
    if (col_ids) bytes += align8((size_t)n_cols * 4);
    std::vector<size_t> row_off(buckets.size(), 0), id_off(buckets.size(), 0), pre_off(buckets.size(), 0);
    std::vector<char> is_contig(buckets.size(), 0);
    std::vector<std::vector<int32_t>> tri_prefix(buckets.size()), diag_prefix(buckets.size());
    std::vector<size_t> diag_off(buckets.size(), 0);
    std::vector<int> refs_per_wg(buckets.size(), 0);
    for (size_t b = 0; b < buckets.size(); ++b) {
        const long col_blocks = call_col_blocks;
        refs_per_wg[b] = lcsgpu::refs_per_block_for(buckets[b].bv, buckets[b].quirk, (long)buckets[b].items.size(), col_blocks, fuse != nullptr);
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
            if (fuse && fuse->prune && acc > 0) {
                // the same workgroups by diagonals (RowsArgs::diag_prefix): rows with more than d column blocks are a prefix
                // of the order above (fullest first), so the tiles d blocks from their row's end are diag[d] .. diag[d + 1]
                const int widest = pre[1] - pre[0];
                std::vector<int32_t>& dg = diag_prefix[b];
                dg.assign((size_t)widest + 1, 0);
                for (int k = 0; k < gy; ++k) {
                    const int cols_k = pre[k + 1] - pre[k];
                    if (cols_k > 0) dg[(size_t)cols_k - 1] += 1; // rows whose LAST diagonal index is cols_k - 1
                }
                // rows with more than d blocks = sum over e >= d of dg[e]; prefix over d of that count
                int64_t rows_with_more = 0;
                std::vector<int32_t> cnt((size_t)widest, 0);
                for (int d = widest - 1; d >= 0; --d) {
                    rows_with_more += dg[(size_t)d];
                    cnt[(size_t)d] = (int32_t)rows_with_more;
                }
                int64_t at = 0;
                for (int d = 0; d < widest; ++d) {
                    dg[(size_t)d] = (int32_t)at;
                    at += cnt[(size_t)d];
                }
                dg[(size_t)widest] = (int32_t)at; // == acc
                diag_off[b] = bytes;
                bytes += align8(dg.size() * 4);
            }
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
            if (!diag_prefix[b].empty()) memcpy(h + diag_off[b], diag_prefix[b].data(), diag_prefix[b].size() * 4);
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

    const hipStream_t run_stream = L.stream;
    HIP_TRY(hipEventRecord(L.ev_start, run_stream));
    // (a call whose refs fall into several half-word classes is several launches, one after the other on the lane's stream)
    for (size_t b = 0; b < buckets.size(); ++b) {
        const Bucket& bk = buckets[b];
        hipStream_t st = run_stream;
        RowsArgs a{};
        a.tiles = (const uint8_t*)ctx->d_tiles.p;
        a.tile_base = (const uint64_t*)ctx->d_tile_base.p;
        a.lens = (const uint32_t*)ctx->d_lens.p;
        a.masks = (const uint64_t*)ctx->d_masks.p;
        a.mask_base = (const uint64_t*)ctx->d_mask_base.p;
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
        if (fuse) {
            a.fuse = *fuse;
            a.fuse.on = 1;
        }
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
            if (!diag_prefix[b].empty()) {
                a.diag_prefix = (const int32_t*)((char*)L.d_plan.p + diag_off[b]);
                a.diag_count = (int32_t)diag_prefix[b].size() - 1;
            }
            const int total = tri_prefix[b].back();
            if (total > 0) {
                HIP_TRY(lcsgpu::launch_rows(bk.bv, bk.quirk, a, total, 1, st));
                ++L.last_launches;
            }
        } else if (bk.bv != 0) {
            if (gy > 65535) return fail(LCSGPU_E_INVALID, "too many ref tiles in one call (%d)", gy);
            HIP_TRY(lcsgpu::launch_rows(bk.bv, bk.quirk, a, gx, gy, st));
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
    HIP_TRY(hipEventRecord(L.ev_stop, run_stream));
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
    g_last.also.clear();
    g_last.pending_on_lane0 = false;
    g_last.ms = f;
    g_last.launches = L.last_launches;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->total_kernel_ms += f;
}


void note_host_call(lcsgpu_ctx* ctx, double ms, int launches)
{
    g_last.ctx = ctx;
    g_last.also.clear();
    g_last.pending_on_lane0 = false;
    g_last.ms = ms;
    g_last.launches = launches;
}

void note_async_call(lcsgpu_ctx* ctx, bool also_this)
{
    if (also_this) {
        g_last.also.push_back(ctx);
        return;
    }
    g_last.ctx = ctx;
    g_last.also.clear();
    g_last.pending_on_lane0 = true;
}

} // namespace lcsgpu_impl

using namespace lcsgpu_impl;

extern "C" {


const char* lcsgpu_version(void)
{
    // both LCS translation units went through the register pass and its check, or the string says what did not
    static const std::string a = lcsgpu::recolor_state(), b = lcsgpu::recolor_state_fused();
    static const std::string v = std::string("lcsgpu 0.4 gfx950 recolor=") + (a == b ? a : (a == "failed" || b == "failed") ? "failed" : "off") +
                                 " kernels=" + lcsgpu::kernel_id() + "/" + lcsgpu::kernel_id_fused();
    return v.c_str();
}
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
    if (const char* e = getenv("LCSGPU_LANES")) n_lanes = std::max(1, std::min(MAX_LANES, atoi(e)));
    // One hardware queue per lane up to 16, so that the lanes' small kernels really overlap: the runtime's
    // default of 4 queues makes 16 streams share them and serialises their launches (1 M-sequence
    // MedoidTree: 4.7 s -> 2.6 s of tree stage).  More than 16 queues cost more than they give (3 x 10^6-sequence
    // MedoidTree, tree stage: 16 queues 2.6 s, 24: 3.0 s, 32: 3.5 s, 64: 7.2 s), further lanes share the 16.  Read by
    // the HIP runtime when it initialises, i.e. effective if this is the process's first HIP call; an explicit
    // setting by the user wins.
    {
        char buf[16];
        snprintf(buf, sizeof buf, "%d", std::min(n_lanes, 16));
        setenv("GPU_MAX_HW_QUEUES", buf, 0);
    }
    const bool profile = getenv("LCSGPU_PROFILE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(LCSGPU_E_NODEVICE, "no HIP device available (%s)", hipGetErrorString(e));
    const double t1 = now();
    if (device_id < 0 || device_id >= n)
        return fail(LCSGPU_E_INVALID, "device %d not in [0,%d)", device_id, n);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(LCSGPU_E_NODEVICE, "device %d is %s; this library is built for gfx950 only", device_id,
                    prop.gcnArchName);
    const double t2 = now();
    HIP_TRY(hipSetDevice(device_id));
    lcsgpu_ctx* ctx = new (std::nothrow) lcsgpu_ctx;
    if (!ctx) return fail(LCSGPU_E_NOMEM, "out of host memory");
    ctx->device = device_id;
    ctx->lanes.resize(LANE_SLOTS); // (beyond MAX_LANES: LaneGuard::FRONT)
    ctx->lane_limit = n_lanes;
    // lane 0 now; the others (and the streams of the CLARANS batches) when first needed -- see Lane::created
    if (!create_lane(ctx, ctx->lanes[0])) {
        lcsgpu_destroy(ctx);
        return fail(LCSGPU_E_HIP, "stream/event creation failed");
    }
    ctx->lanes[0].created = true;
    if (profile)
        fprintf(stderr, "lcsgpu_create: HIP runtime start (hipGetDeviceCount) %.3f s, device properties %.3f s, context + first lane %.3f s\n",
                t1 - t0, t2 - t1, now() - t2);
    *out_ctx = ctx;
    return LCSGPU_OK;
}

int lcsgpu_reserve_lanes(lcsgpu_ctx* ctx, int32_t n_threads)
{
    if (!ctx) return fail(LCSGPU_E_INVALID, "NULL ctx");
    int limit;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        ctx->lane_limit = std::max(ctx->lane_limit, std::min<int>(MAX_LANES, n_threads + 1));
        limit = ctx->lane_limit; // other threads may raise it meanwhile: this call works with what it saw under the lock
    }
    // the first 16 now, the others when a call first needs them (created by the calling threads, in parallel)
    for (size_t i = 1; i < (size_t)limit && (int32_t)i <= std::min(n_threads, 16); ++i) {
        Lane& l = ctx->lanes[i];
        {
            std::lock_guard<std::mutex> lk(ctx->mu);
            if (l.created || l.unusable || l.busy) continue;
            l.busy = true; // reserved while it is being created
        }
        const bool ok = create_lane(ctx, l);
        {
            std::lock_guard<std::mutex> lk(ctx->mu);
            l.created = ok;
            l.unusable = !ok;
            l.busy = false;
        }
        ctx->cv.notify_all();
    }
    if (n_threads > 1) { LaneGuard front(ctx, LaneGuard::FRONT); } // a level-by-level caller: its front lane
    return LCSGPU_OK;
}

int lcsgpu_destroy(lcsgpu_ctx* ctx)
{
    if (!ctx) return LCSGPU_OK;
    (void)hipSetDevice(ctx->device);
    text_release(ctx);
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
        if (l.copy_stream) (void)hipStreamDestroy(l.copy_stream);
        if (l.stream) (void)hipStreamDestroy(l.stream);
    }
    for (DevBuf& c : ctx->sample_chunks) c.release();
    ctx->sample_chunks.clear();
    ctx->d_tiles.release();
    ctx->d_tile_base.release();
    ctx->d_lens.release();
    ctx->d_minlen.release();
    ctx->d_masks.release();
    ctx->d_mask_base.release();
    ctx->d_pow.release();
    ctx->d_powf.release();
    ctx->d_prim.release();
    ctx->d_mst.release();
    ctx->d_gather.release();
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
    struct Lut {
        uint8_t v[256];
        Lut()
        {
            static const char alphabet[25] = "ARNDCQEGHILKMFPSTWYVBZX*";
            for (int ch = 0; ch < 256; ++ch) {
                char c = (char)ch;
                if (c > 'Z') c = (char)(c - 32);
                uint8_t code = 22;
                for (int i = 0; i < 25; ++i)
                    if (alphabet[i] == c) {
                        code = (uint8_t)i;
                        break;
                    }
                v[ch] = code;
            }
        }
    };
    static const Lut table; // built once (the FASTA reader calls this per line)
    const uint8_t* lut = table.v;
    size_t m = 0;
    for (size_t i = 0; i < n; ++i)
        if (residues[i] != '-') codes[m++] = lut[(unsigned char)residues[i]];
    *n_codes = m;
    return LCSGPU_OK;
}

int lcsgpu_upload(lcsgpu_ctx* ctx, const uint8_t* codes, const uint64_t* offsets, int32_t n)
{
    return lcsgpu_upload_ordered(ctx, codes, offsets, n, nullptr, n);
}

int lcsgpu_upload_ordered(lcsgpu_ctx* ctx, const uint8_t* codes, const uint64_t* offsets, int32_t n_records, const int32_t* order,
                          int32_t n)
{
    if (!ctx || !offsets || n < 0 || n_records < 0 || (!order && n != n_records) || (!codes && n_records > 0 && offsets[n_records] > 0))
        return fail(LCSGPU_E_INVALID, "bad argument");
    if (n > 0 && n_records == 0) return fail(LCSGPU_E_INVALID, "an order over no records");
    if (n > 0 && offsets[0] != 0) return fail(LCSGPU_E_INVALID, "offsets[0] must be 0");
    auto record = [&](int32_t i) { return order ? order[i] : i; };
    LaneGuard guard(ctx, LaneGuard::ALL); // nothing may run while the set is replaced
    HIP_TRY(hipSetDevice(ctx->device));
    for (Lane& l : ctx->lanes)
        if (l.created) HIP_TRY(hipStreamSynchronize(l.stream));
    ctx->n = -1;
    ctx->mst.active = false;
    text_release(ctx); // row blocks of the previous set
    for (DevBuf& c : ctx->sample_chunks) c.release();
    ctx->sample_chunks.clear();
    std::vector<uint32_t> lens(n);
    std::vector<uint8_t> quirk(n);
    uint32_t max_len = 0;
    for (int32_t i = 0; i < n; ++i) {
        const int32_t r = record(i);
        if (r < 0 || r >= n_records) return fail(LCSGPU_E_INVALID, "order[%d] = %d is not a record", i, r);
        if (offsets[r + 1] < offsets[r] || offsets[r + 1] > offsets[n_records])
            return fail(LCSGPU_E_INVALID, "offsets not monotone at %d", r);
        const uint64_t len = offsets[r + 1] - offsets[r];
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
    std::vector<uint64_t> mask_base((size_t)n + 1, 0); // rows of 32 x u64 per 64-residue word
    for (int32_t i = 0; i < n; ++i) mask_base[i + 1] = mask_base[i] + (lens[i] + 63) / 64;
    const size_t total = (size_t)tile_base[n_tiles];
    const size_t raw_bytes = n ? (size_t)(offsets[n_records] - offsets[0]) : 0;
    hipError_t e = hipSuccess;
    if ((e = ctx->d_tiles.reserve(std::max<size_t>(total, 16))) != hipSuccess ||
        (e = ctx->d_tile_base.reserve(((size_t)n_tiles + 1) * 8)) != hipSuccess ||
        (e = ctx->d_lens.reserve(std::max<size_t>((size_t)n * 4, 16))) != hipSuccess ||
        (e = ctx->d_mask_base.reserve(((size_t)n + 1) * 8)) != hipSuccess ||
        (e = ctx->d_masks.reserve(std::max<size_t>((size_t)mask_base[n] * 256, 256))) != hipSuccess)
        return fail(LCSGPU_E_NOMEM, "device allocation failed: %s", hipGetErrorString(e));
    HIP_TRY(hipMemcpy(ctx->d_mask_base.p, mask_base.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ctx->d_tile_base.p, tile_base.data(), ((size_t)n_tiles + 1) * 8, hipMemcpyHostToDevice));
    if (n) HIP_TRY(hipMemcpy(ctx->d_lens.p, lens.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    {   // block minima of the lengths: the integer pre-filter of the per-row minima / MST passes (mst_kernels.hip)
        const size_t n16 = ((size_t)n + 15) / 16, n1k = ((size_t)n + 1023) / 1024;
        std::vector<uint32_t> ml(n16 + n1k + 1, ~0u);
        for (int32_t i = 0; i < n; ++i) {
            ml[(size_t)i / 16] = std::min(ml[(size_t)i / 16], lens[i]);
            ml[n16 + (size_t)i / 1024] = std::min(ml[n16 + (size_t)i / 1024], lens[i]);
        }
        HIP_TRY(ctx->d_minlen.reserve(ml.size() * 4));
        HIP_TRY(hipMemcpy(ctx->d_minlen.p, ml.data(), ml.size() * 4, hipMemcpyHostToDevice));
    }
    if (n) {
        DevBuf d_raw, d_off, d_quirk, d_order; // only needed while the tiles are built
        struct Release {
            DevBuf &a, &b, &c, &d;
            ~Release() { a.release(); b.release(); c.release(); d.release(); }
        } release{d_raw, d_off, d_quirk, d_order};
        if ((e = d_raw.reserve(std::max<size_t>(raw_bytes, 16))) != hipSuccess ||
            (e = d_off.reserve(((size_t)n_records + 1) * 8)) != hipSuccess || (e = d_quirk.reserve((size_t)n + 16)) != hipSuccess ||
            (order && (e = d_order.reserve((size_t)n * 4)) != hipSuccess))
            return fail(LCSGPU_E_NOMEM, "device allocation failed: %s", hipGetErrorString(e));
        hipStream_t st = ctx->lanes[0].stream;
        if (raw_bytes) HIP_TRY(hipMemcpyAsync(d_raw.p, codes, raw_bytes, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d_off.p, offsets, ((size_t)n_records + 1) * 8, hipMemcpyHostToDevice, st));
        if (order) HIP_TRY(hipMemcpyAsync(d_order.p, order, (size_t)n * 4, hipMemcpyHostToDevice, st));
        int32_t* d_flags = (int32_t*)((char*)d_quirk.p + (((size_t)n + 3) & ~(size_t)3));
        HIP_TRY(hipMemsetAsync(d_flags, 0, 4, st));
        HIP_TRY(lcsgpu::launch_build_set((const uint8_t*)d_raw.p, (const uint64_t*)d_off.p, order ? (const int32_t*)d_order.p : nullptr,
                                         (const uint64_t*)ctx->d_tile_base.p, n, (uint8_t*)ctx->d_tiles.p,
                                         (uint8_t*)d_quirk.p, d_flags, (const uint64_t*)ctx->d_mask_base.p,
                                         (uint64_t*)ctx->d_masks.p, st));
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
    ctx->ref_class.resize((size_t)n);
    for (int32_t i = 0; i < n; ++i) ctx->ref_class[(size_t)i] = (uint8_t)(lcsgpu::h_class(ctx->lens[(size_t)i]) | (ctx->quirk[(size_t)i] ? 0x80 : 0));
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
    const int64_t off = (int64_t)row_begin * (row_begin - 1) / 2;
    if ((size_t)count * elem_size >= ((size_t)256 << 20) && row_end - row_begin >= 64) {
        // A large triangle leaves in row slices of equal pair counts: slice k is copied to the caller's
        // buffer (its own stream) while slice k+1 is computed -- at 100 000 x 400 aa the 10 GB of results
        // cost 0.57 s of transfer after a 1.41 s kernel when done one after the other.
        const int n_slices = 8;
        std::vector<int32_t> cut(n_slices + 1, row_end);
        cut[0] = row_begin;
        for (int k = 1; k < n_slices; ++k) {
            const double target = (double)off + (double)count * k / n_slices; // pairs below the cut
            int32_t r = (int32_t)std::floor(0.5 + std::sqrt(0.25 + 2.0 * target));
            cut[k] = std::min(row_end, std::max(cut[k - 1], r));
        }
        std::vector<hipEvent_t> done(n_slices, nullptr);
        struct Events {
            std::vector<hipEvent_t>& v;
            ~Events() { for (hipEvent_t e : v) if (e) (void)hipEventDestroy(e); }
        } events{done};
        double ms = 0;
        int launches = 0;
        if (!L.copy_stream) HIP_TRY(hipStreamCreateWithFlags(&L.copy_stream, hipStreamNonBlocking));
        auto copy_slice = [&](int k) -> int {
            const int64_t a0 = (int64_t)cut[k] * (cut[k] - 1) / 2 - off, a1 = (int64_t)cut[k + 1] * (cut[k + 1] - 1) / 2 - off;
            if (a1 <= a0) return LCSGPU_OK;
            HIP_TRY(hipStreamWaitEvent(L.copy_stream, done[k], 0));
            HIP_TRY(hipMemcpyAsync((char*)out + a0 * elem_size, (char*)L.d_out.p + a0 * elem_size, (size_t)(a1 - a0) * elem_size,
                                   hipMemcpyDeviceToHost, L.copy_stream));
            return LCSGPU_OK;
        };
        for (int k = 0; k < n_slices; ++k) {
            if (cut[k + 1] > cut[k]) {
                int rc = run_rows(ctx, L, lcsgpu::MODE_TRIANGLE, nullptr, cut[k], cut[k + 1] - cut[k], nullptr, 0,
                                  std::max(0, cut[k + 1] - 1), L.d_out.p, 0, off, elem_size, cut[k]);
                if (rc) return rc;
            }
            HIP_TRY(hipEventCreateWithFlags(&done[k], hipEventDisableTiming));
            HIP_TRY(hipEventRecord(done[k], L.stream));
            if (k > 0) {
                int rc = copy_slice(k - 1); // returns when the slice is in the caller's (pageable) buffer
                if (rc) return rc;
                // slice k-1's kernel has finished (its results were just copied): its timing is final
            }
            if (cut[k + 1] > cut[k]) {
                HIP_TRY(hipEventSynchronize(L.ev_stop)); // kernel k; the copy above overlapped with it
                finish_host_call(ctx, L);
                ms += g_last.ms;
                launches += g_last.launches;
            }
        }
        int rc = copy_slice(n_slices - 1);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(L.copy_stream));
        g_last.ms = ms;
        g_last.launches = launches;
        return LCSGPU_OK;
    }
    int rc = triangle_common(ctx, L, row_begin, row_end, L.d_out.p, elem_size);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, L.d_out.p, (size_t)count * elem_size, hipMemcpyDeviceToHost, L.stream));
    HIP_TRY(hipEventRecord(L.ev_done, L.stream));
    HIP_TRY(hipEventSynchronize(L.ev_done)); // sleeps; worker threads must not burn a core per pending call
    finish_host_call(ctx, L);
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
    const bool further = std::find(g_last.also.begin(), g_last.also.end(), ctx) != g_last.also.end();
    if (g_last.ctx != ctx && !further) return LCSGPU_OK;
    if (!g_last.pending_on_lane0 && !further) { // a completed host-memory call of this thread
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
