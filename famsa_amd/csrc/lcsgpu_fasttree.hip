// lcsgpu_fasttree.hip -- C-ABI entry points for the MedoidTree / PartTree recursion: the leaf matrices
// of a split in one call, the seed assignment of an evaluation, CLARANS (kernels in
// clarans_kernels.hip; here the host side: the two mt19937 streams and the batch of searches).
#include "lcsgpu_internal.h"
#include "fasttree_kernels.h"

#include <functional>
#include <memory>

using namespace lcsgpu_impl;
using lcsgpu::RowsArgs;

// The packed LCS triangles of several id lists into lane L's result buffer (device memory), list g at pair offset
// tri_base[g]: validation, planning and the launches of lcsgpu_lcs_triangles_batch.  *count = pairs in total (0: nothing to do).  On return the
// launches are queued on L.stream (or, with a ref beyond 2048 residues in the batch, already finished).
// `before_launch` (may be empty) is called once the plan is made, before the first thing goes to the device: the caller's moment
// to take the compute gate (planning a few hundred lists is tens of milliseconds of host work nobody should wait for).
static int batch_triangles_to_device(lcsgpu_ctx* ctx, Lane& L, const int32_t* ids, const int64_t* group_offsets, int32_t n_groups,
                                     int elem_size, std::vector<int64_t>& tri_base, int64_t* count_out, bool* had_long,
                                     const std::function<void()>& before_launch = nullptr)
{
    if (!ctx) return fail(LCSGPU_E_INVALID, "NULL ctx");
    if (ctx->n < 0) return fail(LCSGPU_E_STATE, "no sequence set uploaded");
    if (n_groups < 0 || (n_groups > 0 && !group_offsets)) return fail(LCSGPU_E_INVALID, "bad group table");
    if (elem_size != 2 && elem_size != 4) return fail(LCSGPU_E_INVALID, "elem_size must be 2 or 4");
    if (elem_size == 2 && ctx->max_len > 65535)
        return fail(LCSGPU_E_INVALID, "uint16 output needs all sequences <= 65535 residues");
    *count_out = 0;
    *had_long = false;
    if (n_groups == 0) return LCSGPU_OK;
    if (group_offsets[0] != 0) return fail(LCSGPU_E_INVALID, "group_offsets[0] must be 0");
    const int64_t n_total = group_offsets[n_groups];
    if (n_total < 0 || n_total > 0x7fffffff) return fail(LCSGPU_E_INVALID, "bad total id count");
    tri_base.assign((size_t)n_groups + 1, 0);
    bool any_long = false;
    for (int32_t g = 0; g < n_groups; ++g) {
        const int64_t m = group_offsets[g + 1] - group_offsets[g];
        if (m < 0) return fail(LCSGPU_E_INVALID, "group_offsets not ascending");
        tri_base[g + 1] = tri_base[g] + m * (m - 1) / 2;
    }
    const int64_t count = tri_base[n_groups];
    if (count <= 0) return LCSGPU_OK;
    if (!ids) return fail(LCSGPU_E_INVALID, "NULL ids");
    std::vector<uint8_t> cls((size_t)n_total); // lcsgpu_ctx::ref_class of every id: the one pass over the set's tables
    for (int64_t p = 0; p < n_total; ++p) {
        if (ids[p] < 0 || ids[p] >= ctx->n) return fail(LCSGPU_E_INVALID, "id %d out of range", ids[p]);
        cls[(size_t)p] = ctx->ref_class[(size_t)ids[p]];
        any_long |= (cls[(size_t)p] & 0x7f) == 0;
    }
    *count_out = count;
    *had_long = any_long;
    HIP_TRY(hipSetDevice(ctx->device));
    static std::atomic<long> plan_us[4], plan_calls{0}; // LCSGPU_PROFILE
    auto t_mark = std::chrono::steady_clock::now();
    auto lap = [&](int what) {
        const auto t = std::chrono::steady_clock::now();
        plan_us[what] += std::chrono::duration_cast<std::chrono::microseconds>(t - t_mark).count();
        t_mark = t;
    };
    // (a lane that serves these calls serves many of them, of the caller's batch size give or take: one allocation each)
    HIP_TRY(L.d_out.reserve(L.d_out.cap ? (size_t)count * elem_size : (size_t)count * elem_size * 3 / 2));
    lap(0);
    if (any_long) { // the long-ref kernel keeps its 2-D grid: list by list
        if (before_launch) before_launch();
        double ms = 0;
        int launches = 0;
        for (int32_t g = 0; g < n_groups; ++g) {
            const int32_t m = (int32_t)(group_offsets[g + 1] - group_offsets[g]);
            if (m < 2) continue;
            const int32_t* gi = ids + group_offsets[g];
            int rc = run_rows(ctx, L, lcsgpu::MODE_TRIANGLE, gi, 0, m, gi, 0, m - 1,
                              (char*)L.d_out.p + (size_t)tri_base[g] * elem_size, 0, 0, elem_size);
            if (rc) return rc;
            HIP_TRY(hipStreamSynchronize(L.stream));
            finish_host_call(ctx, L);
            ms += g_last.ms;
            launches += g_last.launches;
        }
        g_last.ms = ms;
        g_last.launches = launches;
        return LCSGPU_OK;
    }

    // per instantiated kernel: its refs (in position order, so the refs of one list are adjacent) and its jobs
    struct BatchBucket {
        int bv;
        bool quirk;
        bool narrow; // one-wave workgroups, column blocks of 64: the refs of short lists (RowsArgs::block_threads)
        std::vector<int32_t> ref_id, ref_col0;
        std::vector<int64_t> ref_row, ref_out0;
        std::vector<int32_t> ref_group;
        std::vector<int4> jobs;
        int refs_per_wg = 0;
    };
    std::vector<BatchBucket> buckets;
    int index_of[320];
    std::fill(index_of, index_of + 320, -1);
    static const int narrow_max = tune_int("narrow_lists", 192); // lists up to that many members run in one-wave workgroups
    int target[65];
    {   // small neighbouring half-word classes share a launch (merge_small_classes, lcsgpu_api.hip)
        double wgs[65] = {0}, inv_refs[65];
        for (int h = 0; h <= 64; ++h) inv_refs[h] = 1.0 / lcsgpu::refs_per_block_for(h, false, 1, 1);
        for (int32_t g = 0; g < n_groups; ++g)
            for (int64_t p = group_offsets[g] + 1; p < group_offsets[g + 1]; ++p) {
                if (cls[(size_t)p] & 0x80) continue;
                const int h = cls[(size_t)p];
                wgs[h] += (double)((p - group_offsets[g] + 255) / 256) * inv_refs[h];
            }
        merge_small_classes(wgs, target);
    }
    for (int32_t g = 0; g < n_groups; ++g)
        for (int64_t p = group_offsets[g]; p < group_offsets[g + 1]; ++p) {
            if (p == group_offsets[g]) continue; // the first member of a list has no partner
            const int32_t id = ids[p];
            const bool q = (cls[(size_t)p] & 0x80) != 0;
            const int bv = q ? lcsgpu::quirk_h_class(ctx->lens[id]) : target[cls[(size_t)p]];
            const bool narrow = !q && group_offsets[g + 1] - group_offsets[g] <= narrow_max;
            const int key = bv * 2 + (q ? 1 : 0) + (narrow ? 160 : 0);
            if (index_of[key] < 0) {
                index_of[key] = (int)buckets.size();
                buckets.push_back(BatchBucket{bv, q, narrow, {}, {}, {}, {}, {}, {}, 0});
                BatchBucket& nb = buckets.back(); // (hundreds of lists: growing these step by step was a third of the planning time)
                const size_t room = (size_t)n_total / (buckets.size() > 1 ? 4 : 1) + 16;
                nb.ref_id.reserve(room);
                nb.ref_row.reserve(room);
                nb.ref_col0.reserve(room);
                nb.ref_out0.reserve(room);
                nb.ref_group.reserve(room);
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
        b.jobs.reserve(nr_all / (size_t)std::max(1, R) * (b.narrow ? 3 : 5) + 16);
        for (size_t k0 = 0; k0 < nr_all;) {
            size_t k1 = k0 + 1;
            while (k1 < nr_all && k1 - k0 < (size_t)R && b.ref_group[k1] == b.ref_group[k0]) ++k1;
            const int32_t g0 = b.ref_col0[k0];
            const int32_t max_row = (int32_t)b.ref_row[k1 - 1]; // rows ascend inside a list
            for (int32_t c0 = g0; c0 < max_row; c0 += b.narrow ? 64 : 256)
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
    lap(1);
    if (L.plan_in_flight) {
        HIP_TRY(hipStreamSynchronize(L.stream));
        L.plan_in_flight = false;
    }
    HIP_TRY(L.h_plan.reserve(L.h_plan.cap ? bytes : bytes * 3 / 2));
    HIP_TRY(L.d_plan.reserve(L.d_plan.cap ? bytes : bytes * 3 / 2));
    lap(2);
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
    lap(3);
    if (getenv("LCSGPU_PROFILE") && (++plan_calls % 20) == 0)
        fprintf(stderr, "triangle batches planned: %ld, thread-ms each: result buffer %.1f, lists + jobs %.1f, plan buffers %.1f, filling them %.1f (last: %lld ids, %d lists)\n",
                plan_calls.load(), 1e-3 * plan_us[0] / plan_calls, 1e-3 * plan_us[1] / plan_calls, 1e-3 * plan_us[2] / plan_calls, 1e-3 * plan_us[3] / plan_calls,
                (long long)n_total, n_groups);
    if (before_launch) before_launch();
    HIP_TRY(hipMemcpyAsync(L.d_plan.p, h, bytes, hipMemcpyHostToDevice, L.stream));
    L.plan_in_flight = true;
    L.last_launches = 0;
    const hipStream_t run_stream = L.stream;
    HIP_TRY(hipEventRecord(L.ev_start, run_stream));
    for (size_t bi = 0; bi < buckets.size(); ++bi) {
        const BatchBucket& b = buckets[bi];
        if (b.jobs.empty()) continue;
        RowsArgs a{};
        a.tiles = (const uint8_t*)ctx->d_tiles.p;
        a.tile_base = (const uint64_t*)ctx->d_tile_base.p;
        a.lens = (const uint32_t*)ctx->d_lens.p;
        a.masks = (const uint64_t*)ctx->d_masks.p;
        a.mask_base = (const uint64_t*)ctx->d_mask_base.p;
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
        a.block_threads = b.narrow ? 64 : 0;
        HIP_TRY(lcsgpu::launch_rows(b.bv, b.quirk, a, (int)b.jobs.size(), 1, run_stream));
        ++L.last_launches;
    }
    HIP_TRY(hipEventRecord(L.ev_stop, run_stream));
    L.timing_valid = true;
    return LCSGPU_OK;
}

extern "C" {

int lcsgpu_lcs_triangles_batch(lcsgpu_ctx* ctx, const int32_t* ids, const int64_t* group_offsets, int32_t n_groups,
                               void* out, int elem_size)
{
    if (!ctx) return fail(LCSGPU_E_INVALID, "NULL ctx");
    // LCSGPU_PROFILE: where the calls' time goes, summed over the calling threads (printed by the 64th, 128th, ... call)
    static std::atomic<long> prof_us[5], prof_calls{0};
    auto t_mark = std::chrono::steady_clock::now();
    auto lap = [&](int what) {
        const auto t = std::chrono::steady_clock::now();
        prof_us[what] += std::chrono::duration_cast<std::chrono::microseconds>(t - t_mark).count();
        t_mark = t;
    };
    LaneGuard guard(ctx, LaneGuard::ANY);
    Lane& L = guard.lane();
    lap(0);
    std::vector<int64_t> tri_base;
    int64_t count = 0;
    bool had_long = false;
    int rc;
    {
        struct Shared { // from the first launch until the kernels have run; the plan is made before, the results travel after
            ComputeGate& g;
            bool held = false;
            ~Shared() { if (held) g.unlock_shared(); }
        } hold{ctx->gate};
        rc = batch_triangles_to_device(ctx, L, ids, group_offsets, n_groups, elem_size, tri_base, &count, &had_long, [&] {
            lap(1);
            hold.g.lock_shared();
            hold.held = true;
            lap(2);
        });
        if (rc || count <= 0) return rc;
        if (!had_long) {
            HIP_TRY(hipEventRecord(L.ev_done, L.stream));
            HIP_TRY(hipEventSynchronize(L.ev_done));
        }
        lap(3);
    }
    if (!out) return fail(LCSGPU_E_INVALID, "NULL out");
    if (had_long) { // every list was synchronised already; the timing is in g_last
        HIP_TRY(hipMemcpy(out, L.d_out.p, (size_t)count * elem_size, hipMemcpyDeviceToHost));
        return LCSGPU_OK;
    }
    HIP_TRY(hipMemcpyAsync(out, L.d_out.p, (size_t)count * elem_size, hipMemcpyDeviceToHost, L.stream));
    HIP_TRY(hipEventRecord(L.ev_done, L.stream));
    HIP_TRY(hipEventSynchronize(L.ev_done));
    finish_host_call(ctx, L);
    lap(4);
    if (getenv("LCSGPU_PROFILE") && (++prof_calls % 19) == 0)
        fprintf(stderr, "triangles_batch: %ld calls so far, thread-ms per call: lane %.1f, plan %.1f, gate %.1f, kernels %.1f, results to the host %.1f\n",
                prof_calls.load(), 1e-3 * prof_us[0] / prof_calls, 1e-3 * prof_us[1] / prof_calls, 1e-3 * prof_us[2] / prof_calls,
                1e-3 * prof_us[3] / prof_calls, 1e-3 * prof_us[4] / prof_calls);
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

int lcsgpu_assign_seeds_batch(lcsgpu_ctx* ctx, const int32_t* seed_ids, const int64_t* seed_offsets, const int32_t* col_ids,
                              const int64_t* col_offsets, int32_t n_jobs, int distance_kind, float* dist, int32_t* assign)
{
    if (!ctx) return fail(LCSGPU_E_INVALID, "NULL ctx");
    if (ctx->n < 0) return fail(LCSGPU_E_STATE, "no sequence set uploaded");
    if (distance_kind != LCSGPU_DIST_INDEL_DIV_LCS && distance_kind != LCSGPU_DIST_INDEL075_DIV_LCS)
        return fail(LCSGPU_E_INVALID, "unknown distance kind %d", distance_kind);
    if (n_jobs < 0 || (n_jobs > 0 && (!seed_offsets || !col_offsets))) return fail(LCSGPU_E_INVALID, "bad job tables");
    if (n_jobs == 0) return LCSGPU_OK;
    if (seed_offsets[0] != 0 || col_offsets[0] != 0) return fail(LCSGPU_E_INVALID, "offsets must start at 0");
    for (int32_t j = 0; j < n_jobs; ++j)
        if (seed_offsets[j + 1] < seed_offsets[j] || col_offsets[j + 1] < col_offsets[j] || seed_offsets[j + 1] - seed_offsets[j] > 0x7fffffff)
            return fail(LCSGPU_E_INVALID, "offsets not ascending at job %d", j);
    const int64_t n_seeds_all = seed_offsets[n_jobs], n_cols_all = col_offsets[n_jobs];
    if (n_cols_all == 0) return LCSGPU_OK;
    if (!col_ids || !dist || !assign || (n_seeds_all > 0 && !seed_ids)) return fail(LCSGPU_E_INVALID, "NULL argument");
    if (n_seeds_all > 0x7fffffff) return fail(LCSGPU_E_INVALID, "too many seeds");
    bool any_long = false;
    for (int64_t s = 0; s < n_seeds_all; ++s) {
        if (seed_ids[s] < 0 || seed_ids[s] >= ctx->n) return fail(LCSGPU_E_INVALID, "seed id %d out of range", seed_ids[s]);
        any_long |= ctx->lens[seed_ids[s]] > 2048;
    }
    for (int64_t c = 0; c < n_cols_all; ++c)
        if (col_ids[c] < 0 || col_ids[c] >= ctx->n) return fail(LCSGPU_E_INVALID, "col id %d out of range", col_ids[c]);
    if (any_long) { // the long-ref kernel keeps its 2-D grid: evaluation by evaluation
        for (int32_t j = 0; j < n_jobs; ++j) {
            const int64_t c0 = col_offsets[j], nc = col_offsets[j + 1] - c0;
            if (nc > 0x7fffffff) return fail(LCSGPU_E_INVALID, "too many columns in job %d", j);
            for (int64_t c = 0; c < nc; ++c) {
                dist[c0 + c] = std::numeric_limits<float>::infinity();
                assign[c0 + c] = 0;
            }
            int rc = lcsgpu_assign_seeds(ctx, seed_ids + seed_offsets[j], (int32_t)(seed_offsets[j + 1] - seed_offsets[j]), col_ids + c0, (int32_t)nc,
                                         distance_kind, 0, dist + c0, assign + c0);
            if (rc) return rc;
        }
        return LCSGPU_OK;
    }
    LaneGuard guard(ctx, LaneGuard::FRONT);
    Lane& L = guard.lane();
    HIP_TRY(hipSetDevice(ctx->device));
    const int elem = ctx->max_len > 65535 ? 4 : 2;
    // pieces = (job, column range), taken in order, so that a launch covers one contiguous range of the caller's columns;
    // a launch's rectangles stay below `cap` bytes
    const int64_t cap = (int64_t)std::max(1, tune_int("assign_batch_kb", 2048 << 10)) << 10;
    struct Piece { int32_t job; int64_t c0, c1; };
    std::vector<Piece> pieces;
    for (int32_t j = 0; j < n_jobs; ++j) {
        const int64_t ns = seed_offsets[j + 1] - seed_offsets[j], c0 = col_offsets[j], c1 = col_offsets[j + 1];
        if (c1 == c0) continue;
        if (ns == 0) { // no seed: nothing beats +inf
            for (int64_t c = c0; c < c1; ++c) { dist[c] = std::numeric_limits<float>::infinity(); assign[c] = 0; }
            continue;
        }
        const int64_t step = std::max<int64_t>(256, (cap / (ns * elem)) & ~(int64_t)255);
        for (int64_t a = c0; a < c1; a += step) pieces.push_back(Piece{j, a, std::min(c1, a + step)});
    }
    HIP_TRY(L.d_draws.reserve((size_t)n_seeds_all * 4 + 16)); // (the lane's spare buffer: the seeds of all jobs)
    HIP_TRY(hipMemcpyAsync(L.d_draws.p, seed_ids, (size_t)n_seeds_all * 4, hipMemcpyHostToDevice, L.stream));
    double ms = 0;
    int launches = 0;
    auto a256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
    for (size_t p0 = 0; p0 < pieces.size();) {
        // the launch: pieces [p0, p1) as far as their columns are contiguous and the rectangles fit
        size_t p1 = p0;
        int64_t bytes = 0;
        while (p1 < pieces.size()) {
            const Piece& pc = pieces[p1];
            const int64_t ns = seed_offsets[pc.job + 1] - seed_offsets[pc.job];
            const int64_t b = ns * (pc.c1 - pc.c0) * elem;
            if (p1 > p0 && (bytes + b > cap || pieces[p1 - 1].c1 != pc.c0 || pc.c1 - pieces[p0].c0 > 0x7fffff00)) break;
            bytes += b;
            ++p1;
        }
        const int64_t C0 = pieces[p0].c0, C1 = pieces[p1 - 1].c1, nc = C1 - C0;
        if (nc > 0x7fffff00) return fail(LCSGPU_E_INVALID, "a job's column range is too long for one launch");
        const int32_t np = (int32_t)(p1 - p0);
        // per instantiated kernel: its refs and jobs
        struct Bucket {
            int bv;
            bool quirk;
            std::vector<int32_t> ref_id, ref_col0, ref_piece;
            std::vector<int64_t> ref_out0;
            std::vector<int4> jobs;
            int refs_per_wg = 0;
        };
        std::vector<Bucket> buckets;
        int index_of[160];
        std::fill(index_of, index_of + 160, -1);
        std::vector<lcsgpu::AssignPiece> table((size_t)np);
        int target[65];
        {
            double wgs[65] = {0};
            for (size_t p = p0; p < p1; ++p) {
                const Piece& pc = pieces[p];
                for (int64_t s = seed_offsets[pc.job]; s < seed_offsets[pc.job + 1]; ++s) {
                    if (ctx->quirk[seed_ids[s]]) continue;
                    const int h = lcsgpu::h_class(ctx->lens[seed_ids[s]]);
                    wgs[h] += (double)((pc.c1 - pc.c0 + 255) / 256) / lcsgpu::refs_per_block_for(h, false, 1, 1);
                }
            }
            merge_small_classes(wgs, target);
        }
        int64_t out_at = 0;
        for (size_t p = p0; p < p1; ++p) {
            const Piece& pc = pieces[p];
            const int64_t s0 = seed_offsets[pc.job], ns = seed_offsets[pc.job + 1] - s0, w = pc.c1 - pc.c0;
            table[p - p0] = lcsgpu::AssignPiece{out_at, (int32_t)(pc.c0 - C0), (int32_t)w, (int32_t)s0, (int32_t)ns};
            for (int64_t r = 0; r < ns; ++r) {
                const int32_t id = seed_ids[s0 + r];
                const bool q = ctx->quirk[id] != 0;
                const int bv = q ? lcsgpu::quirk_h_class(ctx->lens[id]) : target[lcsgpu::h_class(ctx->lens[id])];
                const int key = bv * 2 + (q ? 1 : 0);
                if (index_of[key] < 0) {
                    index_of[key] = (int)buckets.size();
                    buckets.push_back(Bucket{bv, q, {}, {}, {}, {}, {}, 0});
                }
                Bucket& b = buckets[index_of[key]];
                b.ref_id.push_back(id);
                b.ref_col0.push_back((int32_t)(pc.c0 - C0));
                b.ref_piece.push_back((int32_t)(p - p0));
                b.ref_out0.push_back(out_at + r * w);
            }
            out_at += ns * w;
        }
        size_t plan = 0;
        const size_t o_table = plan; plan += a256((size_t)np * sizeof(lcsgpu::AssignPiece));
        std::vector<size_t> o_id(buckets.size()), o_c0(buckets.size()), o_out(buckets.size()), o_job(buckets.size());
        for (size_t bi = 0; bi < buckets.size(); ++bi) {
            Bucket& b = buckets[bi];
            const size_t nr = b.ref_id.size();
            const int R = b.refs_per_wg = lcsgpu::refs_per_block_for(b.bv, b.quirk, (long)nr, (long)std::max<int64_t>(1, nc / 256 / std::max(1, np)));
            for (size_t k0 = 0; k0 < nr;) {
                size_t k1 = k0 + 1;
                while (k1 < nr && k1 - k0 < (size_t)R && b.ref_piece[k1] == b.ref_piece[k0]) ++k1;
                const lcsgpu::AssignPiece& tp = table[(size_t)b.ref_piece[k0]];
                for (int32_t c = tp.col0; c < tp.col0 + tp.n_cols; c += 256) b.jobs.push_back(make_int4((int)k0, (int)(k1 - k0), c, tp.col0 + tp.n_cols));
                k0 = k1;
            }
            if (b.jobs.size() > 0x7fffffffu) return fail(LCSGPU_E_INVALID, "batch too large");
            o_id[bi] = plan; plan += a256(nr * 4);
            o_c0[bi] = plan; plan += a256(nr * 4);
            o_out[bi] = plan; plan += a256(nr * 8);
            o_job[bi] = plan; plan += a256(b.jobs.size() * sizeof(int4));
        }
        if (L.plan_in_flight) {
            HIP_TRY(hipStreamSynchronize(L.stream));
            L.plan_in_flight = false;
        }
        HIP_TRY(L.h_plan.reserve(plan));
        HIP_TRY(L.d_plan.reserve(plan));
        char* h = (char*)L.h_plan.p;
        memcpy(h + o_table, table.data(), (size_t)np * sizeof(lcsgpu::AssignPiece));
        for (size_t bi = 0; bi < buckets.size(); ++bi) {
            const Bucket& b = buckets[bi];
            memcpy(h + o_id[bi], b.ref_id.data(), b.ref_id.size() * 4);
            memcpy(h + o_c0[bi], b.ref_col0.data(), b.ref_col0.size() * 4);
            memcpy(h + o_out[bi], b.ref_out0.data(), b.ref_out0.size() * 8);
            memcpy(h + o_job[bi], b.jobs.data(), b.jobs.size() * sizeof(int4));
        }
        HIP_TRY(hipMemcpyAsync(L.d_plan.p, h, plan, hipMemcpyHostToDevice, L.stream));
        L.plan_in_flight = true;
        // columns, distances, assignments of the launch
        const size_t o_cols = 0, o_dist = o_cols + a256((size_t)nc * 4), o_assign = o_dist + a256((size_t)nc * 4), work = o_assign + a256((size_t)nc * 4);
        HIP_TRY(L.d_work.reserve(work));
        int rc = reserve_big(ctx, L.d_out, (size_t)out_at * elem, "assign_seeds_batch rectangles");
        if (rc) return rc;
        char* base = (char*)L.d_work.p;
        HIP_TRY(hipMemcpyAsync(base + o_cols, col_ids + C0, (size_t)nc * 4, hipMemcpyHostToDevice, L.stream));
        L.last_launches = 0;
        HIP_TRY(hipEventRecord(L.ev_start, L.stream));
        for (size_t bi = 0; bi < buckets.size(); ++bi) {
            const Bucket& b = buckets[bi];
            if (b.jobs.empty()) continue;
            RowsArgs a{};
            a.tiles = (const uint8_t*)ctx->d_tiles.p;
            a.tile_base = (const uint64_t*)ctx->d_tile_base.p;
            a.lens = (const uint32_t*)ctx->d_lens.p;
            a.masks = (const uint64_t*)ctx->d_masks.p;
            a.mask_base = (const uint64_t*)ctx->d_mask_base.p;
            a.n_refs = (int32_t)b.ref_id.size();
            char* d = (char*)L.d_plan.p;
            a.ref_ids = (const int32_t*)(d + o_id[bi]);
            a.ref_rows = nullptr;
            a.ref_col0 = (const int32_t*)(d + o_c0[bi]);
            a.ref_out0 = (const int64_t*)(d + o_out[bi]);
            a.jobs = (const int4*)(d + o_job[bi]);
            a.col_ids = (const int32_t*)(base + o_cols);
            a.n_cols = (int32_t)nc;
            a.out = L.d_out.p;
            a.elem_size = elem;
            a.mode = lcsgpu::MODE_RECT;
            a.refs_per_block = b.refs_per_wg;
            HIP_TRY(lcsgpu::launch_rows(b.bv, b.quirk, a, (int)b.jobs.size(), 1, L.stream));
            ++L.last_launches;
        }
        HIP_TRY(hipEventRecord(L.ev_stop, L.stream));
        L.timing_valid = true;
        HIP_TRY(lcsgpu::launch_assign_seeds_batch(L.d_out.p, elem, (const lcsgpu::AssignPiece*)((char*)L.d_plan.p + o_table), np,
                                                  (const int32_t*)L.d_draws.p, (const int32_t*)(base + o_cols), nc, (const uint32_t*)ctx->d_lens.p,
                                                  (const float*)ctx->d_powf.p, distance_kind, (float*)(base + o_dist), (int32_t*)(base + o_assign), L.stream));
        HIP_TRY(hipMemcpyAsync(dist + C0, base + o_dist, (size_t)nc * 4, hipMemcpyDeviceToHost, L.stream));
        HIP_TRY(hipMemcpyAsync(assign + C0, base + o_assign, (size_t)nc * 4, hipMemcpyDeviceToHost, L.stream));
        HIP_TRY(hipEventRecord(L.ev_done, L.stream));
        HIP_TRY(hipEventSynchronize(L.ev_done));
        finish_host_call(ctx, L);
        ms += g_last.ms;
        launches += g_last.launches;
        p0 = p1;
    }
    g_last.ms = ms;
    g_last.launches = launches;
    return LCSGPU_OK;
}

// CLARANS for many samples in one go: all sample triangles in one batched LCS launch, all float matrices in one launch,
// then ONE workgroup per sample runs that sample's whole chain of local searches (clarans_chain_kernel); the host only
// pre-draws the step positions and composes the shuffles (neither generator looks at a search, Clustering.cpp:43-46) --
// both depend on the shape (members, medoids) alone, so samples of one shape share them.
int lcsgpu_clarans_batch(lcsgpu_ctx* ctx, const int32_t* ids, const int64_t* offsets, int32_t n_jobs, int distance_kind,
                         const int32_t* n_medoids, int32_t n_fixed, float explore_fraction, int32_t num_local, int32_t* medoids_out)
{
    if (!ctx) return fail(LCSGPU_E_INVALID, "NULL ctx");
    if (ctx->n < 0) return fail(LCSGPU_E_STATE, "no sequence set uploaded");
    if (distance_kind != LCSGPU_DIST_INDEL_DIV_LCS && distance_kind != LCSGPU_DIST_INDEL075_DIV_LCS)
        return fail(LCSGPU_E_INVALID, "unknown distance kind %d", distance_kind);
    if (n_jobs < 0 || num_local < 1 || n_fixed < 0) return fail(LCSGPU_E_INVALID, "bad CLARANS batch");
    if (n_jobs == 0) return LCSGPU_OK;
    if (!ids || !offsets || !n_medoids || !medoids_out) return fail(LCSGPU_E_INVALID, "NULL argument");
    if (offsets[0] != 0) return fail(LCSGPU_E_INVALID, "offsets[0] must be 0");
    std::vector<int64_t> med_off((size_t)n_jobs + 1, 0);
    for (int32_t j = 0; j < n_jobs; ++j) {
        const int64_t n = offsets[j + 1] - offsets[j];
        const int32_t k = n_medoids[j];
        if (n < 1 || n > 0x7fffffff) return fail(LCSGPU_E_INVALID, "bad sample %d", j);
        if (k < 1 || k > n || n_fixed >= k)
            return fail(LCSGPU_E_INVALID, "bad CLARANS shape: %d medoids (%d fixed) of %lld, %d searches", k, n_fixed, (long long)n, num_local);
        if (k > lcsgpu::CLARANS_MAX_MEDOIDS || n - k > lcsgpu::CLARANS_MAX_NONMEDOIDS)
            return fail(LCSGPU_E_UNSUPPORTED, "device CLARANS handles at most %d medoids and %d other sample members (asked: %d, %lld)",
                        lcsgpu::CLARANS_MAX_MEDOIDS, lcsgpu::CLARANS_MAX_NONMEDOIDS, k, (long long)(n - k));
        med_off[j + 1] = med_off[j] + k;
    }
    for (int64_t i = 0; i < offsets[n_jobs]; ++i)
        if (ids[i] < 0 || ids[i] >= ctx->n) return fail(LCSGPU_E_INVALID, "sample id %d out of range", ids[i]);

    // what depends on a sample's shape alone: the composed shuffles and the step positions
    struct Shape {
        int32_t n, k;
        std::vector<int32_t> perm; // [num_local][n]
        std::mt19937 gen_positions;
        std::vector<int32_t> draws;
        DevBuf d_perm, d_draws;
        int corrected = 0;
    };
    std::vector<std::unique_ptr<Shape>> shapes;
    struct Release {
        std::vector<std::unique_ptr<Shape>>& v;
        ~Release() { for (auto& s : v) { s->d_perm.release(); s->d_draws.release(); } }
    } release{shapes};
    std::vector<int> shape_of((size_t)n_jobs, -1);
    std::vector<int32_t> live; // the samples that need a search
    for (int32_t j = 0; j < n_jobs; ++j) {
        const int32_t n = (int32_t)(offsets[j + 1] - offsets[j]), k = n_medoids[j];
        int si = -1;
        for (size_t t = 0; t < shapes.size(); ++t)
            if (shapes[t]->n == n && shapes[t]->k == k) si = (int)t;
        if (si < 0) {
            shapes.emplace_back(new Shape);
            Shape& sh = *shapes.back();
            sh.n = n;
            sh.k = k;
            // Clustering.cpp:21-29: how many non-improving steps end a local search
            const int n_swaps = (n - k) * k, min_max_neighbor = 250;
            const int max_neighbor = n_swaps < min_max_neighbor ? n_swaps : std::max((int)(explore_fraction * n_swaps), min_max_neighbor);
            sh.corrected = max_neighbor / k;
            // partial_shuffle(candidate + n_fixed, candidate + n, candidate + n, gen_nodes) before every local search
            // (deterministic_random.h:113-127), applied to positions: where[i] = the position whose content ends up at i
            std::mt19937 gen_nodes;
            sh.perm.resize((size_t)num_local * n);
            std::vector<int32_t> where((size_t)n);
            for (int it = 0; it < num_local; ++it) {
                for (int32_t i = 0; i < n; ++i) where[(size_t)i] = i;
                int32_t* first = where.data() + n_fixed;
                const long cnt = n - n_fixed, N = cnt - 1;
                for (long i = 0; i < cnt; ++i) {
                    const unsigned long d = (unsigned long)N - (unsigned long)i + 1;
                    const unsigned long r = (unsigned long)gen_nodes(); // < 2^32: never in the rejected tail of a 64-bit range
                    std::swap(first[i], first[(r % d) + (unsigned long)i]);
                }
                std::copy(where.begin(), where.end(), sh.perm.begin() + (size_t)it * n);
            }
            si = (int)shapes.size() - 1;
        }
        shape_of[(size_t)j] = si;
        if (n == k) {
            // every member is a medoid: no step is ever drawn, every search costs 0, the first one stands
            // (Clustering.cpp:239-257: a later search has to be cheaper) -- its medoids are the order after the first shuffle
            std::copy(shapes[(size_t)si]->perm.begin(), shapes[(size_t)si]->perm.begin() + n, medoids_out + med_off[j]);
        } else
            live.push_back(j);
    }
    if (live.empty()) return LCSGPU_OK;

    const auto t_call = std::chrono::steady_clock::now();
    LaneGuard guard(ctx, LaneGuard::FRONT);
    Lane& L = guard.lane();
    HIP_TRY(hipSetDevice(ctx->device));
    const bool profile = getenv("LCSGPU_PROFILE") != nullptr;
    double t_prof[6] = {0};
    auto t_mark = std::chrono::steady_clock::now();
    auto lap = [&](int what, bool sync) {
        if (sync) (void)hipStreamSynchronize(L.stream);
        if (!profile) return;
        const auto t = std::chrono::steady_clock::now();
        t_prof[what] += std::chrono::duration<double>(t - t_mark).count();
        t_mark = t;
    };
    std::unique_lock<ComputeGate> wave(ctx->gate, std::defer_lock);
    const int elem = ctx->max_len > 65535 ? 4 : 2;
    std::vector<int64_t> tri_base;
    int64_t count = 0;
    bool had_long = false;
    int rc = batch_triangles_to_device(ctx, L, ids, offsets, n_jobs, elem, tri_base, &count, &had_long, [&] { lap(1, false); });
    if (rc) return rc;
    lap(2, profile);
    double lcs_ms = 0;
    int lcs_launches = 0;
    if (had_long) { // (synchronised list by list: the timing is in g_last)
        lcs_ms = g_last.ms;
        lcs_launches = g_last.launches;
    }
    // the samples' buffers, one allocation
    auto a256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const int n_live = (int)live.size();
    const int64_t n_total = offsets[n_jobs];
    size_t at = 0;
    const size_t o_chains = at; at += a256((size_t)n_live * sizeof(lcsgpu::ClaransChain));
    const size_t o_states = at; at += a256((size_t)n_live * lcsgpu::CLARANS_STATE_WORDS * 4);
    const size_t o_ids = at; at += a256((size_t)n_total * 4);
    const size_t o_best = at; at += a256((size_t)med_off[n_jobs] * 4);
    struct Place { char* D; char* DM; char* cand; char* st; char* log; };
    std::vector<Place> place((size_t)n_live);
    int max_n = 0, max_k = 0;
    size_t sample_bytes = 0;
    {   // every sample's work area out of the context's chunks (this call owns them: it holds the FRONT lane); what the chunks
        // there are cannot hold comes out of ONE new chunk
        auto area = [&](int q, size_t* o_DM, size_t* o_cand, size_t* o_st, size_t* o_log) {
            const int32_t j = live[(size_t)q];
            const size_t n = (size_t)(offsets[j + 1] - offsets[j]), k = (size_t)n_medoids[j];
            *o_DM = a256(n * n * 4);
            *o_cand = *o_DM + a256(n * k * 4);
            *o_st = *o_cand + a256(n * 4);
            *o_log = *o_st + a256(n * 16);
            return *o_log + a256((n + 1) * 4);
        };
        size_t chunk = 0, used = 0; // the chunk being filled
        for (int q = 0; q < n_live; ++q) {
            const int32_t j = live[(size_t)q];
            const size_t n = (size_t)(offsets[j + 1] - offsets[j]), k = (size_t)n_medoids[j];
            size_t o_DM, o_cand, o_st, o_log;
            const size_t o_D = 0, need = area(q, &o_DM, &o_cand, &o_st, &o_log);
            while (chunk < ctx->sample_chunks.size() && ctx->sample_chunks[chunk].cap - used < need) {
                ++chunk;
                used = 0;
            }
            if (chunk == ctx->sample_chunks.size()) {
                ctx->sample_chunks.emplace_back();
                size_t rest = 0, a, b, c, d;
                for (int r = q; r < n_live; ++r) rest += area(r, &a, &b, &c, &d);
                rc = reserve_big(ctx, ctx->sample_chunks.back(), rest, "CLARANS samples");
                if (rc) {
                    ctx->sample_chunks.pop_back();
                    return rc;
                }
                used = 0;
            }
            char* b = (char*)ctx->sample_chunks[chunk].p + used;
            place[(size_t)q] = Place{b + o_D, b + o_DM, b + o_cand, b + o_st, b + o_log};
            used += need;
            sample_bytes += need;
            max_n = std::max(max_n, (int)n);
            max_k = std::max(max_k, (int)k);
        }
    }
    rc = reserve_big(ctx, L.d_work, at, "CLARANS batch tables");
    if (rc) return rc;
    at += sample_bytes; // (the profile line's figure)
    lap(3, false);
    char* base = (char*)L.d_work.p;
    const size_t host_bytes = a256((size_t)n_live * sizeof(lcsgpu::ClaransChain)) + a256((size_t)n_live * lcsgpu::CLARANS_STATE_WORDS * 4);
    HIP_TRY(L.h_small.reserve(host_bytes));
    lcsgpu::ClaransChain* h_chains = (lcsgpu::ClaransChain*)L.h_small.p;
    int32_t* h_states = (int32_t*)((char*)L.h_small.p + a256((size_t)n_live * sizeof(lcsgpu::ClaransChain)));
    static const int draws_first = std::max(16, tune_int("clarans_draws", 65536)), slice_us = std::max(0, tune_int("clarans_slice_us", 0));
    for (auto& shp : shapes) {
        Shape& sh = *shp;
        if (sh.n == sh.k) continue;
        HIP_TRY(sh.d_perm.reserve(sh.perm.size() * 4));
        HIP_TRY(hipMemcpyAsync(sh.d_perm.p, sh.perm.data(), sh.perm.size() * 4, hipMemcpyHostToDevice, L.stream));
    }
    // det_uniform_int_distribution<int>(k, n - 1) over gen_positions (deterministic_random.h:62-76), `want` of them
    auto extend_draws = [&](Shape& sh, size_t want) -> int {
        if (sh.draws.size() >= want) return LCSGPU_OK;
        const uint32_t k = (uint32_t)sh.k, diff = (uint32_t)(sh.n - sh.k);
        const uint32_t bad = 0xffffffffu / diff;
        sh.draws.reserve(want);
        while (sh.draws.size() < want) {
            const uint32_t r = sh.gen_positions();
            if (r / diff < bad) sh.draws.push_back((int32_t)(r % diff + k));
        }
        HIP_TRY(sh.d_draws.reserve(want * 4)); // (only between launches: nothing reads the old array)
        HIP_TRY(hipMemcpyAsync(sh.d_draws.p, sh.draws.data(), sh.draws.size() * 4, hipMemcpyHostToDevice, L.stream));
        return LCSGPU_OK;
    };
    for (auto& shp : shapes)
        if (shp->n != shp->k) {
            rc = extend_draws(*shp, (size_t)draws_first);
            if (rc) return rc;
        }
    HIP_TRY(hipMemcpyAsync(base + o_ids, ids, (size_t)n_total * 4, hipMemcpyHostToDevice, L.stream));
    memset(h_states, 0, (size_t)n_live * lcsgpu::CLARANS_STATE_WORDS * 4);
    const float flt_max = std::numeric_limits<float>::max();
    for (int q = 0; q < n_live; ++q) {
        const int32_t j = live[(size_t)q];
        const Shape& sh = *shapes[(size_t)shape_of[(size_t)j]];
        lcsgpu::ClaransChain& c = h_chains[q];
        memset(&c, 0, sizeof c);
        const Place& p = place[(size_t)q];
        c.a.D = (const float*)p.D;
        c.a.DMt = (float*)p.DM;
        c.a.cand = (int32_t*)p.cand;
        c.a.st = (float4*)p.st;
        c.a.cost_log = (float*)p.log;
        c.a.state = (int32_t*)(base + o_states) + (size_t)q * lcsgpu::CLARANS_STATE_WORDS;
        c.a.n_elems = sh.n;
        c.a.n_medoids = sh.k;
        c.a.n_fixed = n_fixed;
        c.a.corrected = sh.corrected;
        c.a.draws = (const int32_t*)sh.d_draws.p;
        c.a.draws_len = (int32_t)sh.draws.size();
        c.perm = (const int32_t*)sh.d_perm.p;
        c.ids = (const int32_t*)(base + o_ids) + offsets[j];
        c.best = (int32_t*)(base + o_best) + med_off[j];
        c.tri0 = tri_base[(size_t)j];
        c.num_local = num_local;
        int32_t* st = h_states + (size_t)q * lcsgpu::CLARANS_STATE_WORDS;
        st[4] = 1;  // ST_FRESH
        st[10] = 1; // ST_FIRST
        memcpy(&st[17], &flt_max, 4); // ST_BEST
        st[18] = 1; // ST_NEED_INIT
    }
    HIP_TRY(hipMemcpyAsync(base + o_states, h_states, (size_t)n_live * lcsgpu::CLARANS_STATE_WORDS * 4, hipMemcpyHostToDevice, L.stream));
    HIP_TRY(hipMemcpyAsync(base + o_chains, h_chains, (size_t)n_live * sizeof(lcsgpu::ClaransChain), hipMemcpyHostToDevice, L.stream));
    HIP_TRY(lcsgpu::launch_subset_distances_batch(L.d_out.p, elem, (const lcsgpu::ClaransChain*)(base + o_chains), n_live, max_n,
                                                  (const uint32_t*)ctx->d_lens.p, (const float*)ctx->d_powf.p, distance_kind, L.stream));
    // the triangles and distances ran beside the other lanes' work (this lane's stream goes first); the chains run alone:
    // the chip to them (lcsgpu_internal.h, ComputeGate)
    HIP_TRY(hipEventRecord(L.ev_done, L.stream));
    HIP_TRY(hipEventSynchronize(L.ev_done));
    lap(4, false);
    wave.lock();
    lap(0, false);
    // the chains: normally one launch; a chain that runs out of positions (or, under LCSGPU_TUNE clarans_slice_us, of time)
    // comes back for another
    std::vector<int> todo((size_t)n_live);
    for (int q = 0; q < n_live; ++q) todo[(size_t)q] = q;
    std::vector<lcsgpu::ClaransChain> all(h_chains, h_chains + n_live);
    int launches = 0;
    long accepts = 0, steps = 0;
    std::vector<int32_t> ticks, where;
    while (!todo.empty()) {
        const int nt = (int)todo.size();
        for (int t = 0; t < nt; ++t) {
            lcsgpu::ClaransChain& c = h_chains[t];
            c = all[(size_t)todo[(size_t)t]];
            const Shape& sh = *shapes[(size_t)shape_of[(size_t)live[(size_t)todo[(size_t)t]]]];
            c.a.draws = (const int32_t*)sh.d_draws.p;
            c.a.draws_len = (int32_t)sh.draws.size();
        }
        HIP_TRY(hipMemcpyAsync(base + o_chains, h_chains, (size_t)nt * sizeof(lcsgpu::ClaransChain), hipMemcpyHostToDevice, L.stream));
        HIP_TRY(lcsgpu::launch_clarans_chains((const lcsgpu::ClaransChain*)(base + o_chains), nt, max_k, slice_us, L.stream));
        HIP_TRY(hipMemcpyAsync(h_states, base + o_states, (size_t)n_live * lcsgpu::CLARANS_STATE_WORDS * 4, hipMemcpyDeviceToHost, L.stream));
        HIP_TRY(hipEventRecord(L.ev_done, L.stream));
        HIP_TRY(hipEventSynchronize(L.ev_done));
        L.plan_in_flight = false;
        ++launches;
        std::vector<int> again;
        for (int q : todo) {
            const int32_t* st = h_states + (size_t)q * lcsgpu::CLARANS_STATE_WORDS;
            if (st[1]) { // ST_DONE: the whole chain
                accepts += st[3];
                steps += st[12];
                ticks.push_back(st[19]);
                where.push_back(st[20]);
                continue;
            }
            again.push_back(q);
            if (st[7]) { // ST_MORE_DRAWS
                Shape& sh = *shapes[(size_t)shape_of[(size_t)live[(size_t)q]]];
                rc = extend_draws(sh, std::max(sh.draws.size() * 2, (size_t)st[0] + (size_t)st[8] + (size_t)draws_first));
                if (rc) return rc;
            }
        }
        todo.swap(again);
        if (launches > 100000) return fail(LCSGPU_E_STATE, "CLARANS batch does not finish");
    }
    // the results: live samples' medoids (the others were written above)
    std::vector<int32_t> best((size_t)med_off[n_jobs]);
    HIP_TRY(hipMemcpyAsync(best.data(), base + o_best, best.size() * 4, hipMemcpyDeviceToHost, L.stream));
    HIP_TRY(hipStreamSynchronize(L.stream));
    for (int32_t j : live) std::copy(best.begin() + med_off[j], best.begin() + med_off[j + 1], medoids_out + med_off[j]);
    if (!had_long) finish_host_call(ctx, L);
    else note_host_call(ctx, lcs_ms, lcs_launches);
    if (getenv("LCSGPU_PROFILE") && !ticks.empty()) {
        std::vector<int32_t> sorted(ticks);
        std::sort(sorted.begin(), sorted.end());
        std::map<int, int> per_cu;
        for (int32_t w : where) ++per_cu[w];
        int shared_cu = 0;
        for (auto& kv : per_cu) shared_cu += kv.second > 1 ? kv.second : 0;
        double slow_shared = 0, slow_alone = 0;
        int n_sh = 0, n_al = 0;
        for (size_t i = 0; i < ticks.size(); ++i) {
            if (per_cu[where[i]] > 1) { slow_shared += ticks[i]; ++n_sh; } else { slow_alone += ticks[i]; ++n_al; }
        }
        fprintf(stderr, "clarans.batch chains: run time min %.1f median %.1f max %.1f ms; %d of %zu shared a CU (mean %.1f ms against %.1f ms alone)\n",
                sorted.front() * 1e-5, sorted[sorted.size() / 2] * 1e-5, sorted.back() * 1e-5, shared_cu, ticks.size(),
                n_sh ? slow_shared / n_sh * 1e-5 : 0.0, n_al ? slow_alone / n_al * 1e-5 : 0.0);
    }
    lap(5, false);
    if (profile)
        fprintf(stderr, "clarans.batch parts: gate %.1f ms, planning the triangles %.1f, their kernels %.1f, work area (%.2f GB) %.1f, tables + distances %.1f, chains + results %.1f\n",
                1e3 * t_prof[0], 1e3 * t_prof[1], 1e3 * t_prof[2], at / 1e9, 1e3 * t_prof[3], 1e3 * t_prof[4], 1e3 * t_prof[5]);
    if (getenv("LCSGPU_PROFILE"))
        fprintf(stderr, "clarans.batch: %d samples (%d searched, %zu shapes), %d launch(es), %ld accepts, %ld steps looked at, %.3f s\n", n_jobs, n_live,
                shapes.size(), launches, accepts, steps, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_call).count());
    return LCSGPU_OK;
}

int lcsgpu_clarans(lcsgpu_ctx* ctx, const int32_t* ids, int32_t n_ids, int distance_kind, int32_t n_medoids,
                   int32_t n_fixed, float explore_fraction, int32_t num_local, int32_t* medoids_out)
{
    if (!ids || !medoids_out || n_ids < 1) return fail(LCSGPU_E_INVALID, "bad sample / output");
    const int64_t offsets[2] = {0, n_ids};
    return lcsgpu_clarans_batch(ctx, ids, offsets, 1, distance_kind, &n_medoids, n_fixed, explore_fraction, num_local, medoids_out);
}

} // extern "C"
