// lcsgpu_trees.hip -- C-ABI entry points of the whole-set tree reducers (Prim, UPGMA, NJ) and the
// per-row minima: the LCS triangle stays in HBM, the kernels of tree_kernels.hip consume it there.
#include "lcsgpu_internal.h"

#include <queue>

using namespace lcsgpu_impl;

extern "C" {

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

    if ((triangle_orientation || nq == 0) && !getenv("LCSGPU_MST_PRIM")) {
        // distances do not depend on which endpoint is the ref: Boruvka rounds over the triangle, then
        // Prim's insertion order from vertex 0 as a walk over the n-1 tree edges
        auto a256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
        const int n_chunks = std::max(1, std::min(32, (n + 1023) / 1024));
        const int rows_per_chunk = (n + n_chunks - 1) / n_chunks;
        const size_t o_comp = 0, o_next = o_comp + a256((size_t)n * 4), o_par = o_next + a256((size_t)n * 4),
                     o_bd = o_par + a256((size_t)n * 4), o_bi = o_bd + a256((size_t)n * 8), o_cd = o_bi + a256((size_t)n * 8),
                     o_ci = o_cd + a256((size_t)n * 8), o_pd = o_ci + a256((size_t)n * 8),
                     o_pi = o_pd + a256((size_t)n_chunks * n * 8), o_edges = o_pi + a256((size_t)n_chunks * n * 8),
                     o_cnt = o_edges + a256((size_t)(n - 1) * sizeof(lcsgpu::MstEdge)), total = o_cnt + 256;
        HIP_TRY(ctx->d_prim.reserve(total));
        char* base = (char*)ctx->d_prim.p;
        lcsgpu::BoruvkaArgs b{};
        b.tri = L.d_out.p;
        b.lens = (const uint32_t*)ctx->d_lens.p;
        b.pow_table = (const double*)ctx->d_pow.p;
        b.comp = (int32_t*)(base + o_comp);
        b.comp_next = (int32_t*)(base + o_next);
        b.parent = (int32_t*)(base + o_par);
        b.best_d = (unsigned long long*)(base + o_bd);
        b.best_id = (unsigned long long*)(base + o_bi);
        b.cb_d = (unsigned long long*)(base + o_cd);
        b.cb_id = (unsigned long long*)(base + o_ci);
        b.part_d = (unsigned long long*)(base + o_pd);
        b.part_id = (unsigned long long*)(base + o_pi);
        b.edges = (lcsgpu::MstEdge*)(base + o_edges);
        b.counters = (int32_t*)(base + o_cnt);
        b.n = n;
        b.kind = distance_kind;
        b.n_chunks = n_chunks;
        b.rows_per_chunk = rows_per_chunk;
        HIP_TRY(lcsgpu::launch_boruvka_init(b, L.stream));
        int32_t found = 0;
        for (int round = 0; found < n - 1; ++round) {
            if (round > 40) return fail(LCSGPU_E_STATE, "MST: the Boruvka rounds do not converge");
            HIP_TRY(lcsgpu::launch_boruvka_round(b, elem, L.stream));
            std::swap(b.comp, b.comp_next);
            const int32_t before = found;
            HIP_TRY(hipMemcpyAsync(&found, b.counters, 4, hipMemcpyDeviceToHost, L.stream));
            HIP_TRY(hipStreamSynchronize(L.stream));
            L.plan_in_flight = false;
            if (found <= before) return fail(LCSGPU_E_STATE, "MST: a Boruvka round added no edge");
        }
        std::vector<lcsgpu::MstEdge> tree((size_t)n - 1);
        HIP_TRY(hipMemcpy(tree.data(), b.edges, tree.size() * sizeof(lcsgpu::MstEdge), hipMemcpyDeviceToHost));
        // Prim from vertex 0 over the tree, edges ordered like MSTPrim's keys: (d, ~pack(min, max))
        std::vector<int32_t> head((size_t)n + 1, 0), adj((size_t)2 * (n - 1));
        for (const auto& e : tree) { ++head[e.from + 1]; ++head[e.to + 1]; }
        for (int32_t v = 0; v < n; ++v) head[v + 1] += head[v];
        {
            std::vector<int32_t> fill(head.begin(), head.end() - 1);
            for (int32_t k = 0; k < n - 1; ++k) { adj[fill[tree[k].from]++] = k; adj[fill[tree[k].to]++] = k; }
        }
        struct Cand {
            double d;
            uint64_t id;
            int32_t edge, to;
            bool operator>(const Cand& o) const { return d > o.d || (d == o.d && id > o.id); }
        };
        std::priority_queue<Cand, std::vector<Cand>, std::greater<Cand>> heap;
        std::vector<char> in_tree(n, 0);
        auto visit = [&](int32_t v) {
            in_tree[v] = 1;
            for (int32_t k = head[v]; k < head[v + 1]; ++k) {
                const lcsgpu::MstEdge& e = tree[adj[k]];
                const int32_t w = e.from == v ? e.to : e.from;
                if (!in_tree[w]) heap.push(Cand{e.dist, ~(((uint64_t)(uint32_t)e.from << 32) + (uint32_t)e.to), adj[k], w});
            }
        };
        visit(0);
        for (int32_t k = 0; k < n - 1; ++k) {
            while (!heap.empty() && in_tree[heap.top().to]) heap.pop();
            if (heap.empty()) return fail(LCSGPU_E_STATE, "MST: the edges found do not span the set");
            const Cand c = heap.top();
            heap.pop();
            out_edges[k].from = tree[c.edge].from;
            out_edges[k].to = tree[c.edge].to;
            out_edges[k].dist = tree[c.edge].dist;
            visit(c.to);
        }
        note_async_call(ctx);
        return LCSGPU_OK;
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
                 o_pd = o_node + a16((size_t)n * 4), o_pj = o_pd + a16((size_t)2 * blocks * 4),
                 o_bd = o_pj + a16((size_t)2 * blocks * 4), o_bj = o_bd + a16((size_t)blocks * 4),
                 o_sel = o_bj + a16((size_t)blocks * 4), o_left = o_sel + 48, o_right = o_left + a16((size_t)n * 4),
                 total = o_right + a16((size_t)n * 4);
    HIP_TRY(ctx->d_prim.reserve(total));
    char* base = (char*)ctx->d_prim.p;
    HIP_TRY(hipMemsetAsync(base + o_sel, 0, 48, L.stream));
    lcsgpu::UpgmaArgs a{};
    a.bm_d = (float*)(base + o_bd);
    a.bm_j = (uint32_t*)(base + o_bj);
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
    uint32_t sel[12] = {0};
    HIP_TRY(hipMemcpyAsync(out_left, a.left, (size_t)(n - 1) * 4, hipMemcpyDeviceToHost, L.stream));
    HIP_TRY(hipMemcpyAsync(out_right, a.right, (size_t)(n - 1) * 4, hipMemcpyDeviceToHost, L.stream));
    HIP_TRY(hipMemcpyAsync(sel, a.sel, 48, hipMemcpyDeviceToHost, L.stream));
    HIP_TRY(hipStreamSynchronize(L.stream));
    L.plan_in_flight = false;
    if (sel[8])
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

} // extern "C"
