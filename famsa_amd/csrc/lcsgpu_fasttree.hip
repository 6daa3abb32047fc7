// lcsgpu_fasttree.hip -- C-ABI entry points for the MedoidTree / PartTree recursion: the leaf matrices
// of a split in one call, the seed assignment of an evaluation, CLARANS (kernels in
// clarans_kernels.hip; here the host side: the two mt19937 streams and the batch of searches).
#include "lcsgpu_internal.h"

using namespace lcsgpu_impl;
using lcsgpu::RowsArgs;

namespace {

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

// One stint as the driver: looks for everything joined, until nothing is left or `mine` is done.
void clarans_drive(lcsgpu_ctx* ctx, ClaransBatcher& B, ClaransJob* mine)
{
    // a look = one launch: every joined search advances for `slice_us` (3 x 10^6 sequences, tree stage: 500 us 1.09-1.11 s,
    // 1000: 1.06-1.14 s, 2000: 1.22-1.30 s) with at least `draws_ahead` pre-drawn positions in front of it
    static const int draws_ahead = std::max(1, tune_int("clarans_draws", 8192)), slice_us = std::max(1, tune_int("clarans_slice_us", 1000));
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
        auto hip_ok = [&](hipError_t e, const char* what) {
            if (e != hipSuccess && rc == LCSGPU_OK) rc = fail(LCSGPU_E_HIP, "%s failed: %s", what, hipGetErrorString(e));
        };
        int32_t* hs = (int32_t*)B.h_states.p;
        // The searches leave their state block in the batch's pinned (device-mapped) buffer themselves: no copy per search
        // and look (7-8 copies of 256 B were 110 us of a 2.4 ms look); without the mapping: a copy each.
        int32_t* hs_dev = nullptr;
        if (hipHostGetDevicePointer((void**)&hs_dev, B.h_states.p, 0) != hipSuccess) {
            (void)hipGetLastError();
            hs_dev = nullptr;
        }
        for (ClaransJob* j : now) {
            // (a search that runs out of positions stops and says so; the next look brings more)
            if (rc == LCSGPU_OK) rc = clarans_extend_draws(*j, (size_t)j->p_host + (size_t)j->state[8] + (size_t)draws_ahead, B.stream);
            lcsgpu::ClaransArgs& s1 = batch.s[batch.n];
            s1 = j->a;
            s1.host_state = hs_dev ? hs_dev + 64 * batch.n : nullptr;
            ++batch.n;
        }
        if (rc == LCSGPU_OK) hip_ok(lcsgpu::launch_clarans_search(batch, slice_us, B.stream), "CLARANS searches");
        if (!hs_dev)
            for (size_t i = 0; i < now.size() && rc == LCSGPU_OK; ++i)
                hip_ok(hipMemcpyAsync(hs + 64 * i, now[i]->a.state, 64, hipMemcpyDeviceToHost, B.stream), "state read-back");
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
                    memcpy(j->state, hs + 64 * i, 64);
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
                    B.prof_searches += 1;
                    B.prof_accepts += j->state[3];
                    B.prof_rounds += j->state[11];
                    B.prof_steps += j->state[12];
                    B.prof_useful += j->state[13];
                    B.prof_no_b += j->state[14];
                    B.prof_no_p += j->state[15];
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
    if (int rc = ensure_batcher(ctx, B)) return rc;
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

} // namespace

extern "C" {

} // extern "C"

// The packed LCS triangles of several id lists into lane L's result buffer (device memory), list g at pair offset
// tri_base[g]: validation, planning and the launches of lcsgpu_lcs_triangles_batch.  *count = pairs in total (0: nothing to do).  On return the
// launches are queued on L.stream (or, with a ref beyond 2048 residues in the batch, already finished).
static int batch_triangles_to_device(lcsgpu_ctx* ctx, Lane& L, const int32_t* ids, const int64_t* group_offsets, int32_t n_groups,
                                     int elem_size, std::vector<int64_t>& tri_base, int64_t* count_out, bool* had_long)
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
    for (int64_t p = 0; p < n_total; ++p) {
        if (ids[p] < 0 || ids[p] >= ctx->n) return fail(LCSGPU_E_INVALID, "id %d out of range", ids[p]);
        any_long |= ctx->lens[ids[p]] > 2048;
    }
    *count_out = count;
    *had_long = any_long;
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
        std::vector<int32_t> ref_id, ref_col0;
        std::vector<int64_t> ref_row, ref_out0;
        std::vector<int32_t> ref_group;
        std::vector<int4> jobs;
        int refs_per_wg = 0;
    };
    std::vector<BatchBucket> buckets;
    int index_of[160];
    std::fill(index_of, index_of + 160, -1);
    int target[65];
    {   // small neighbouring half-word classes share a launch (merge_small_classes, lcsgpu_api.hip)
        double wgs[65] = {0};
        for (int32_t g = 0; g < n_groups; ++g)
            for (int64_t p = group_offsets[g] + 1; p < group_offsets[g + 1]; ++p) {
                const int32_t id = ids[p];
                if (ctx->quirk[id]) continue;
                const int h = lcsgpu::h_class(ctx->lens[id]);
                wgs[h] += (double)((p - group_offsets[g] + 255) / 256) / lcsgpu::refs_per_block_for(h, false, 1, 1);
            }
        merge_small_classes(wgs, target);
    }
    for (int32_t g = 0; g < n_groups; ++g)
        for (int64_t p = group_offsets[g]; p < group_offsets[g + 1]; ++p) {
            if (p == group_offsets[g]) continue; // the first member of a list has no partner
            const int32_t id = ids[p];
            const bool q = ctx->quirk[id] != 0;
            const int bv = q ? lcsgpu::quirk_h_class(ctx->lens[id]) : target[lcsgpu::h_class(ctx->lens[id])];
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
    LaneGuard guard(ctx, LaneGuard::ANY);
    Lane& L = guard.lane();
    std::vector<int64_t> tri_base;
    int64_t count = 0;
    bool had_long = false;
    int rc = batch_triangles_to_device(ctx, L, ids, group_offsets, n_groups, elem_size, tri_base, &count, &had_long);
    if (rc || count <= 0) return rc;
    if (!out) return fail(LCSGPU_E_INVALID, "NULL out");
    if (had_long) { // every list was synchronised already; the timing is in g_last
        HIP_TRY(hipMemcpy(out, L.d_out.p, (size_t)count * elem_size, hipMemcpyDeviceToHost));
        return LCSGPU_OK;
    }
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
    if (n_medoids > lcsgpu::CLARANS_MAX_MEDOIDS || n_ids - n_medoids > lcsgpu::CLARANS_MAX_NONMEDOIDS)
        return fail(LCSGPU_E_UNSUPPORTED, "device CLARANS handles at most %d medoids and %d other sample members (asked: %d, %d)",
                    lcsgpu::CLARANS_MAX_MEDOIDS, lcsgpu::CLARANS_MAX_NONMEDOIDS, n_medoids, n_ids - n_medoids);
    if (n_ids == n_medoids) {
        // every member is a medoid: no step is ever drawn (corrected = 0), every search costs 0, the first one stands
        // (Clustering.cpp:239-257: a later search has to be cheaper) -- its medoids are the order after the first shuffle
        std::mt19937 gen_nodes;
        std::vector<int32_t> cand(n_ids);
        for (int32_t i = 0; i < n_ids; ++i) cand[i] = i;
        int32_t* first = cand.data() + n_fixed;
        const long cnt = n_ids - n_fixed, N = cnt - 1;
        for (long i = 0; i < cnt; ++i) {
            const unsigned long d = (unsigned long)N - (unsigned long)i + 1;
            std::swap(first[i], first[((unsigned long)gen_nodes() % d) + (unsigned long)i]);
        }
        std::copy(cand.begin(), cand.end(), medoids_out);
        return LCSGPU_OK;
    }

    const int32_t n = n_ids, k = n_medoids;
    // Clustering.cpp:21-29: how many non-improving steps end a local search
    const int n_swaps = (n - k) * k;
    const int min_max_neighbor = 250;
    const int max_neighbor = n_swaps < min_max_neighbor
                                 ? n_swaps
                                 : std::max((int)(explore_fraction * n_swaps), min_max_neighbor);
    const int corrected = max_neighbor / k;

    auto t_mark = std::chrono::steady_clock::now();
    auto lap = [&](int what) { // (LCSGPU_PROFILE's account of the call; a few clock readings per call)
        const auto t = std::chrono::steady_clock::now();
        ctx->clarans_us[what] += std::chrono::duration_cast<std::chrono::microseconds>(t - t_mark).count();
        t_mark = t;
    };
    LaneGuard guard(ctx, LaneGuard::ANY);
    Lane& L = guard.lane();
    HIP_TRY(hipSetDevice(ctx->device));
    // everything this call enqueues itself goes to one of the context's high-priority streams, if a multi-threaded caller
    // has announced itself (lcsgpu_ctx::prep_streams); the lane's own stream is idle meanwhile
    struct StreamSwap {
        Lane& l;
        hipStream_t own;
        bool settled = false; // the call has waited for the last thing it queued
        StreamSwap(Lane& lane, hipStream_t s) : l(lane), own(lane.stream) { if (s) l.stream = s; }
        ~StreamSwap()
        {
            // an early return (an error) may leave work of this call queued on the shared stream: the lane and its buffers
            // go back to the pool with this guard, so wait for it
            if (!settled && l.stream != own) (void)hipStreamSynchronize(l.stream);
            l.stream = own;
        }
    } swap(L, ctx->prep_streams[0] ? ctx->prep_streams[ctx->prep_next++ % (ctx->prep_streams[1] ? 2 : 1)] : nullptr);
    const int elem = ctx->max_len > 65535 ? 4 : 2;
    const size_t pairs = (size_t)n * (n - 1) / 2;
    auto a256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_D = 0, o_DM = o_D + a256((size_t)n * n * 4), o_cand = o_DM + a256((size_t)n * k * 4), o_st = o_cand + a256((size_t)n * 4),
                 o_log = o_st + a256((size_t)n * 16), o_state = o_log + a256((size_t)(n + 1) * 4), o_ids = o_state + 256,
                 total = o_ids + a256((size_t)n * 4);
    HIP_TRY(L.d_work.reserve(total));
    HIP_TRY(L.h_small.reserve((size_t)n * 4 + 64));
    char* base = (char*)L.d_work.p;
    if (pairs > 0) HIP_TRY(L.d_out.reserve(pairs * elem));
    lap(0);
    HIP_TRY(hipMemsetAsync(base + o_state, 0, 256, L.stream));
    HIP_TRY(hipMemcpyAsync(base + o_ids, ids, (size_t)n * 4, hipMemcpyHostToDevice, L.stream));
    if (pairs > 0) {
        int rc = run_rows(ctx, L, lcsgpu::MODE_TRIANGLE, ids, 0, n, ids, 0, n - 1, L.d_out.p, 0, 0, elem);
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
        {   // the first draws (the init kernel checks that a window's worth is there)
            int rc = clarans_extend_draws(job, (size_t)job.p_host + (size_t)std::max(corrected, 1), L.stream);
            if (rc) return rc;
        }
        HIP_TRY(lcsgpu::launch_clarans_init(job.a, L.stream));
        HIP_TRY(hipEventRecord(L.ev_done, L.stream));
        HIP_TRY(hipEventSynchronize(L.ev_done)); // the search runs on the batch's stream
        L.plan_in_flight = false;
        lap(iter == 0 ? 1 : 2); // (the first wait of a call also covers the sample's triangle and distances)
        int rc = clarans_run_search(ctx, job);
        if (rc) return rc;
        lap(3);
        float cost;
        memcpy(&cost, &job.state[5], 4);
        HIP_TRY(hipMemcpyAsync(L.h_small.p, a.cand, (size_t)n * 4, hipMemcpyDeviceToHost, L.stream)); // (a blocking hipMemcpy was 0.4 ms under load)
        HIP_TRY(hipEventRecord(L.ev_done, L.stream));
        HIP_TRY(hipEventSynchronize(L.ev_done));
        memcpy(cand.data(), L.h_small.p, (size_t)n * 4);
        lap(4);
        if (cost < best_cost) {
            best_cost = cost;
            std::copy(cand.begin(), cand.begin() + k, medoids_out);
        }
    }
    finish_host_call(ctx, L);
    swap.settled = true;
    ctx->clarans_calls += 1;
    return LCSGPU_OK;
}

} // extern "C"
