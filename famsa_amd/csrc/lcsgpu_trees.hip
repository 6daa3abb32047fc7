// lcsgpu_trees.hip -- C-ABI entry points of the whole-set tree reducers (Prim, UPGMA, NJ) and the
// per-row minima: the LCS triangle stays in HBM, the kernels of tree_kernels.hip consume it there.
#include "lcsgpu_internal.h"
#include "nj_loop.h"

#include <dlfcn.h>

#include <memory>
#include <queue>
#include <thread>

// RCCL: declarations only -- librccl is loaded with dlopen when the exchange of lcsgpu_multi_mst_prim wants it.  A ROCm
// installation without the RCCL development headers still builds the library: the RCCL exchange is then compiled out
// (LCSGPU_EXCHANGE=rccl answers LCSGPU_E_UNSUPPORTED, the automatic choice is the peer-copy exchange).
#if defined(__has_include)
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#define LCSGPU_HAVE_RCCL_HEADERS 1
#endif
#endif
#ifndef LCSGPU_HAVE_RCCL_HEADERS
#define LCSGPU_HAVE_RCCL_HEADERS 0
typedef void* ncclComm_t; // placeholders so that the signatures below compile; no RCCL call is made
#endif

using namespace lcsgpu_impl;

static int64_t tri_offset(int64_t r) { return r * (r - 1) / 2; }

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
                                      ctx->minlen1024(), (const double*)ctx->d_pow.p, distance_kind,
                                      (lcsgpu::RowMin*)d_out, L.stream));
    if (sync) HIP_TRY(hipStreamSynchronize(L.stream));
    return LCSGPU_OK;
}

} // extern "C"

// ---- the sharded MST (Boruvka over row blocks): state set-up shared by the single-context and the
// multi-context entry points.  The caller holds lane 0.
// the block's LCS values with the local half of a Boruvka round folded into the launch (lcs_kernels.h, FuseArgs):
// stored to d_out as well, or -- d_out NULL -- only folded
static int fused_rows(lcsgpu_ctx* ctx, Lane& L, const lcsgpu::BoruvkaArgs& b, void* d_out, int elem, bool with_labels)
{
    // (with labels: a record of the last round whose edge still leaves its vertex's component is this round's best already --
    //  the edges that cross now are a subset of those that crossed then -- and stays; the others start empty)
    HIP_TRY(lcsgpu::launch_boruvka_fuse_reset(b, with_labels, L.stream));
    lcsgpu::FuseArgs f{};
    f.row_rec = b.fuse_row;
    f.col_rec = b.fuse_col;
    f.comp = with_labels ? b.comp : nullptr; // round 0: every vertex is its own component
    f.pow_table = b.pow_table;
    f.kind = b.kind;
    static const int prune = tune_int("mst_length_bound", 1);
    f.prune = prune && !d_out ? 1 : 0; // (a launch that also stores the triangle computes every value)
    f.stats = (unsigned long long*)(b.counters + 8);
    return run_rows(ctx, L, lcsgpu::MODE_TRIANGLE, nullptr, b.r0, b.r1 - b.r0, nullptr, 0, std::max(0, b.r1 - 1), d_out, 0, b.off,
                    elem, b.r0, &f);
}

// d_tri: the block's triangle in device memory, or NULL = none is kept (every round recomputes the block's LCS
// values, O(n) memory).  compute: d_tri is to be FILLED here, by a launch that does round 0's local half as well.
static int shard_begin(lcsgpu_ctx* ctx, Lane& L, void* d_tri, int elem, int32_t r0, int32_t r1, int kind, bool compute = false)
{
    const int32_t n = ctx->n;
    auto a256 = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const int32_t rows = r1 - r0;
    // Row chunks of the column pass.  Its lanes start from the row pass's result, so short streams cost no
    // threshold warm-up any more, and short chunks keep the workgroups that run at the same time on neighbouring
    // rows (the DRAM pages they share): n = 100 000, column passes of one tree: 2 chunks 21.2 ms, 8: 18.2, 16: 16.6,
    // 32: 15.4, 64: 14.1, 128: 13.7 (+0.5 fold), 256: 13.7 (+0.9), 512: 14.2 (+1.8) -> about 1024 rows per chunk
    int n_chunks = std::max(1, std::min(96, (rows + 1023) / 1024));
    {   // a small block has too few column blocks to fill the chip with 1024-row chunks (13 774 sequences: 54 x 14 workgroups,
        // half of them above the diagonal, 62 dependent batches each: 275 us a pass, 0.7 TB/s): shorter chunks until there are
        // ~4096 workgroups, down to 64 rows
        const int col_blocks = std::max(1, (r1 + 255) / 256);
        const int want = std::min({256, (4096 + col_blocks - 1) / col_blocks, std::max(1, rows / 64)});
        n_chunks = std::max(n_chunks, want);
    }
    if (!d_tri) n_chunks = 0; // no passes, no partials
    const int rows_per_chunk = std::max(1, (rows + std::max(n_chunks, 1) - 1) / std::max(n_chunks, 1));
    const size_t key = sizeof(lcsgpu::MstKey);
    const size_t o_comp = 0, o_next = o_comp + a256((size_t)n * 4), o_par = o_next + a256((size_t)n * 4),
                 o_rb = o_par + a256((size_t)n * 4), o_best = o_rb + a256((size_t)n * key),
                 o_vb = o_best + a256((size_t)n * key), o_cd = o_vb + a256((size_t)n * key),
                 o_ci = o_cd + a256((size_t)n * 8), o_part = o_ci + a256((size_t)n * 8),
                 o_edges = o_part + a256((size_t)n_chunks * n * key),
                 o_cnt = o_edges + a256((size_t)std::max(n - 1, 1) * sizeof(lcsgpu::MstEdge)), o_aux = o_cnt + 256,
                 o_frow = o_aux + a256((size_t)n * 8), o_fcol = o_frow + a256((size_t)n * 8),
                 total = o_fcol + a256((size_t)n * 8);
    int rc = reserve_big(ctx, ctx->d_mst, total, "the MST state");
    if (rc) return rc;
    char* base = (char*)ctx->d_mst.p;
    lcsgpu::BoruvkaArgs b{};
    b.tri = d_tri;
    b.off = (int64_t)r0 * (r0 > 0 ? r0 - 1 : 0) / 2;
    b.r0 = r0;
    b.r1 = r1;
    b.lens = (const uint32_t*)ctx->d_lens.p;
    b.pow_table = (const double*)ctx->d_pow.p;
    b.pow_n = (int32_t)std::min<uint64_t>((uint64_t)2 * ctx->max_len + 1, 0x7fffffff);
    b.pow_in_lds = (kind == 1 && (size_t)b.pow_n * sizeof(double) <= 48 * 1024) ? 1 : 0;
    b.comp = (int32_t*)(base + o_comp);
    b.comp_next = (int32_t*)(base + o_next);
    b.parent = (int32_t*)(base + o_par);
    b.row_best = (lcsgpu::MstKey*)(base + o_rb);
    b.best = (lcsgpu::MstKey*)(base + o_best);
    b.vbest = (lcsgpu::MstKey*)(base + o_vb);
    b.cb_d = (unsigned long long*)(base + o_cd);
    b.cb_id = (unsigned long long*)(base + o_ci);
    b.part = (lcsgpu::MstKey*)(base + o_part);
    b.edges = (lcsgpu::MstEdge*)(base + o_edges);
    b.counters = (int32_t*)(base + o_cnt);
    b.row_aux = (uint2*)(base + o_aux);
    b.fuse_row = (unsigned long long*)(base + o_frow);
    b.fuse_col = (unsigned long long*)(base + o_fcol);
    b.minlen16 = ctx->minlen16();
    b.minlen1024 = ctx->minlen1024();
    b.n = n;
    b.kind = kind;
    b.n_chunks = n_chunks;
    b.rows_per_chunk = rows_per_chunk;
    HIP_TRY(lcsgpu::launch_boruvka_init(b, L.stream));
    HIP_TRY(hipMemsetAsync(b.counters + 8, 0, 16, L.stream)); // the fused launches' tile counts (FuseArgs::stats)
    ctx->mst.active = true;
    ctx->mst.b = b;
    ctx->mst.elem = elem;
    ctx->mst.found = 0;
    ctx->mst.rounds = 0;
    ctx->mst.fused_ready = false;
    ctx->mst.passes_stand = false;
    if (d_tri && compute) {
        rc = fused_rows(ctx, L, b, d_tri, elem, false);
        if (rc) {
            ctx->mst.active = false;
            return rc;
        }
        ctx->mst.fused_ready = true;
    }
    return LCSGPU_OK;
}

// local half of a round into d_keys (NULL: the context's own buffer); optionally copied to the host
static int shard_best(lcsgpu_ctx* ctx, Lane& L, void* d_keys, lcsgpu_mst_key* h_keys)
{
    lcsgpu::BoruvkaArgs b = ctx->mst.b;
    if (d_keys) b.best = (lcsgpu::MstKey*)d_keys;
    if (!b.tri) { // nothing resident: this round's LCS values are computed now, the fold fused into the launch
        int rc = fused_rows(ctx, L, b, nullptr, ctx->mst.elem, ctx->mst.rounds > 0);
        if (rc) return rc;
        HIP_TRY(lcsgpu::launch_boruvka_fuse_fold(b, L.stream));
    } else if (ctx->mst.fused_ready && ctx->mst.rounds == 0) {
        HIP_TRY(lcsgpu::launch_boruvka_fuse_fold(b, L.stream)); // the launch that filled the triangle did round 0
        ctx->mst.fused_ready = false;                           // its records are spent: from here on the passes
    } else {
        static const int keep = tune_int("mst_keep", 1), crossmul = tune_int("mst_crossmul", 1);
        b.keep = keep && ctx->mst.passes_stand ? 1 : 0;
        b.crossmul = crossmul ? 1 : 0;
        HIP_TRY(lcsgpu::launch_boruvka_best(b, ctx->mst.elem, L.stream));
        ctx->mst.passes_stand = true;
    }
    if (h_keys) {
        HIP_TRY(hipMemcpyAsync(h_keys, b.best, (size_t)b.n * sizeof(lcsgpu::MstKey), hipMemcpyDeviceToHost, L.stream));
        HIP_TRY(hipStreamSynchronize(L.stream));
        L.plan_in_flight = false;
    }
    return LCSGPU_OK;
}

// global half of a round over n_parts x n gathered keys (NULL: the context's own keys, one part): queued on the
// lane's stream, no host synchronisation
static int shard_merge_async(lcsgpu_ctx* ctx, Lane& L, const void* d_gathered, int32_t n_parts)
{
    lcsgpu::BoruvkaArgs& b = ctx->mst.b;
    if (ctx->mst.rounds > 64) return fail(LCSGPU_E_STATE, "MST: the Boruvka rounds do not converge");
    const lcsgpu::MstKey* g = d_gathered ? (const lcsgpu::MstKey*)d_gathered : b.best;
    HIP_TRY(lcsgpu::launch_boruvka_merge(b, g, d_gathered ? n_parts : 1, L.stream));
    std::swap(b.comp, b.comp_next);
    ++ctx->mst.rounds;
    return LCSGPU_OK;
}

// the number of tree edges recorded so far, once the queued rounds have run (one host synchronisation); the
// merge kernels report a relabelling walk that did not end (inconsistent keys) through counters[1]
static int shard_count(lcsgpu_ctx* ctx, Lane& L, int32_t* n_edges)
{
    lcsgpu::BoruvkaArgs& b = ctx->mst.b;
    int32_t c[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(c, b.counters, 8, hipMemcpyDeviceToHost, L.stream));
    HIP_TRY(hipStreamSynchronize(L.stream));
    L.plan_in_flight = false;
    const int32_t found = c[0];
    if (c[1]) return fail(LCSGPU_E_STATE, "MST: inconsistent keys (a hooking cycle among the components; were the keys gathered from different rounds?)");
    if (found > b.n - 1) return fail(LCSGPU_E_STATE, "MST: %d edges recorded for %d vertices", found, b.n);
    if (found <= ctx->mst.found && found < b.n - 1)
        return fail(LCSGPU_E_STATE, "MST: a Boruvka round added no edge (%d of %d)", found, b.n - 1);
    ctx->mst.found = found;
    if (n_edges) *n_edges = found;
    return LCSGPU_OK;
}

static int shard_merge(lcsgpu_ctx* ctx, Lane& L, const void* d_gathered, int32_t n_parts, int32_t* n_edges)
{
    int rc = shard_merge_async(ctx, L, d_gathered, n_parts);
    return rc ? rc : shard_count(ctx, L, n_edges);
}

static int shard_finish(lcsgpu_ctx* ctx, Lane& L, lcsgpu_mst_edge* out_edges, bool order = true)
{
    const lcsgpu::BoruvkaArgs& b = ctx->mst.b;
    static_assert(sizeof(lcsgpu_mst_edge) == sizeof(lcsgpu::MstEdge), "edge layout");
    if (ctx->mst.found != b.n - 1) return fail(LCSGPU_E_STATE, "MST: %d of %d edges found", ctx->mst.found, b.n - 1);
    HIP_TRY(hipMemcpyAsync(out_edges, b.edges, (size_t)(b.n - 1) * sizeof(lcsgpu_mst_edge), hipMemcpyDeviceToHost, L.stream));
    HIP_TRY(hipStreamSynchronize(L.stream));
    L.plan_in_flight = false;
    return order ? order_edges_like_prim(out_edges, b.n) : LCSGPU_OK;
}

namespace lcsgpu_impl {

// Prim from vertex 0 over the n-1 tree edges, candidates ordered like MSTPrim's keys (d, ~pack(min, max))
// (reference tree/MSTPrim.cpp:372-391): the order in which MSTPrim::run_view adds the edges.
int order_edges_like_prim(lcsgpu_mst_edge* edges, int32_t n)
{
    if (n < 2) return LCSGPU_OK;
    std::vector<lcsgpu_mst_edge> tree(edges, edges + (n - 1));
    for (const auto& e : tree)
        if (e.from < 0 || e.to < 0 || e.from >= n || e.to >= n || e.from >= e.to)
            return fail(LCSGPU_E_INVALID, "MST: edge (%d, %d) is not a pair from < to of [0, %d)", e.from, e.to, n);
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
            const lcsgpu_mst_edge& e = tree[adj[k]];
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
        edges[k] = tree[c.edge];
        visit(c.to);
    }
    return LCSGPU_OK;
}

} // namespace lcsgpu_impl

extern "C" {

static bool valid_kind(int kind) { return kind == LCSGPU_DIST_INDEL_DIV_LCS || kind == LCSGPU_DIST_INDEL075_DIV_LCS; }

int lcsgpu_mst_shard_begin(lcsgpu_ctx* ctx, void* d_triangle, int elem_size, int32_t row_begin, int32_t row_end,
                           int distance_kind)
{
    const bool triangle_orientation = (distance_kind & LCSGPU_MST_TRIANGLE_ORIENTATION) != 0;
    const bool compute = (distance_kind & LCSGPU_MST_COMPUTE) != 0;
    distance_kind &= ~(LCSGPU_MST_TRIANGLE_ORIENTATION | LCSGPU_MST_COMPUTE);
    if (!ctx) return fail(LCSGPU_E_INVALID, "NULL ctx");
    if (ctx->n < 0) return fail(LCSGPU_E_STATE, "no sequence set uploaded");
    if (!valid_kind(distance_kind)) return fail(LCSGPU_E_INVALID, "unknown distance kind %d", distance_kind);
    if (row_begin < 0 || row_end < row_begin || row_end > ctx->n) return fail(LCSGPU_E_INVALID, "bad row range");
    if (elem_size != 2 && elem_size != 4) return fail(LCSGPU_E_INVALID, "elem_size must be 2 or 4");
    if (!d_triangle && !compute && (int64_t)row_end * (row_end - 1) / 2 - (int64_t)row_begin * (row_begin - 1) / 2 > 0)
        return fail(LCSGPU_E_INVALID, "NULL device pointer (pass LCSGPU_MST_COMPUTE to run without a resident triangle)");
    if (compute && ctx->max_len > 65535)
        return fail(LCSGPU_E_UNSUPPORTED, "LCSGPU_MST_COMPUTE needs all sequences <= 65535 residues");
    if (compute && elem_size != 2 && d_triangle) return fail(LCSGPU_E_INVALID, "LCSGPU_MST_COMPUTE fills a uint16 triangle");
    if (!triangle_orientation)
        for (int32_t i = 0; i < ctx->n; ++i)
            if (ctx->quirk[i])
                return fail(LCSGPU_E_UNSUPPORTED, "sequence %d is orientation sensitive: MSTPrim's distances depend on which "
                                                  "endpoint is the ref, the triangle does not hold them (use lcsgpu_mst_prim)", i);
    LaneGuard guard(ctx, LaneGuard::LANE0);
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = shard_begin(ctx, guard.lane(), d_triangle, elem_size, row_begin, row_end, distance_kind, compute);
    if (!rc && compute && d_triangle) note_async_call(ctx); // lcsgpu_last_kernel_ms: the fused LCS launch
    return rc;
}

int lcsgpu_mst_shard_best(lcsgpu_ctx* ctx, void* d_keys, lcsgpu_mst_key* h_keys)
{
    if (!ctx) return fail(LCSGPU_E_INVALID, "NULL ctx");
    if (!ctx->mst.active) return fail(LCSGPU_E_STATE, "lcsgpu_mst_shard_begin has not been called");
    LaneGuard guard(ctx, LaneGuard::LANE0);
    HIP_TRY(hipSetDevice(ctx->device));
    return shard_best(ctx, guard.lane(), d_keys, h_keys);
}

int lcsgpu_mst_shard_merge(lcsgpu_ctx* ctx, const void* d_gathered, int32_t n_parts, int32_t* n_edges)
{
    if (!ctx) return fail(LCSGPU_E_INVALID, "NULL ctx");
    if (!ctx->mst.active) return fail(LCSGPU_E_STATE, "lcsgpu_mst_shard_begin has not been called");
    if (d_gathered && n_parts < 1) return fail(LCSGPU_E_INVALID, "n_parts < 1");
    LaneGuard guard(ctx, LaneGuard::LANE0);
    HIP_TRY(hipSetDevice(ctx->device));
    return shard_merge(ctx, guard.lane(), d_gathered, n_parts, n_edges);
}

int lcsgpu_mst_shard_set_components(lcsgpu_ctx* ctx, const int32_t* comp)
{
    if (!ctx || !comp) return fail(LCSGPU_E_INVALID, "NULL argument");
    if (!ctx->mst.active) return fail(LCSGPU_E_STATE, "lcsgpu_mst_shard_begin has not been called");
    LaneGuard guard(ctx, LaneGuard::LANE0);
    Lane& L = guard.lane();
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipMemcpyAsync(ctx->mst.b.comp, comp, (size_t)ctx->n * 4, hipMemcpyHostToDevice, L.stream));
    HIP_TRY(hipStreamSynchronize(L.stream));
    ctx->mst.passes_stand = false; // the caller's labels need not be a coarsening of the ones the passes saw
    // A round of the host-merge protocol ends here (shard_best -> lcsgpu_mst_merge_host -> this call), as a round of the
    // device protocol ends in shard_merge_async: from now on the labels count (the fused launches fold with them, the
    // records round 0's launch left behind are spent and the passes take over).
    ++ctx->mst.rounds;
    return LCSGPU_OK;
}

int lcsgpu_mst_shard_finish(lcsgpu_ctx* ctx, lcsgpu_mst_edge* out_edges)
{
    if (!ctx) return fail(LCSGPU_E_INVALID, "NULL ctx");
    if (!ctx->mst.active) return fail(LCSGPU_E_STATE, "lcsgpu_mst_shard_begin has not been called");
    if (ctx->n < 2) return LCSGPU_OK;
    if (!out_edges) return fail(LCSGPU_E_INVALID, "NULL out_edges");
    LaneGuard guard(ctx, LaneGuard::LANE0);
    HIP_TRY(hipSetDevice(ctx->device));
    return shard_finish(ctx, guard.lane(), out_edges);
}

int lcsgpu_mst_shard_edges(lcsgpu_ctx* ctx, lcsgpu_mst_edge* out_edges)
{
    if (!ctx) return fail(LCSGPU_E_INVALID, "NULL ctx");
    if (!ctx->mst.active) return fail(LCSGPU_E_STATE, "lcsgpu_mst_shard_begin has not been called");
    if (ctx->n < 2) return LCSGPU_OK;
    if (!out_edges) return fail(LCSGPU_E_INVALID, "NULL out_edges");
    LaneGuard guard(ctx, LaneGuard::LANE0);
    HIP_TRY(hipSetDevice(ctx->device));
    return shard_finish(ctx, guard.lane(), out_edges, false);
}

// Host form of the global half of a round (no GPU involved): the same fold / per-component minimum /
// hooking / relabelling as launch_boruvka_merge, for exchanges that happen in host memory.
int lcsgpu_mst_merge_host(const lcsgpu_mst_key* keys, int32_t n_parts, int32_t n, int32_t* comp, lcsgpu_mst_edge* edges,
                          int32_t* n_edges)
{
    if (!keys || !comp || !edges || !n_edges || n_parts < 1 || n < 0) return fail(LCSGPU_E_INVALID, "bad argument");
    const uint64_t NO_D = 0x7fefffffffffffffull, NO_ID = ~0ull;
    auto less = [](uint64_t d1, uint64_t i1, uint64_t d2, uint64_t i2) { return d1 < d2 || (d1 == d2 && i1 < i2); };
    std::vector<uint64_t> cb_d((size_t)n, NO_D), cb_id((size_t)n, NO_ID);
    for (int32_t v = 0; v < n; ++v) {
        uint64_t bd = NO_D, bi = NO_ID;
        for (int32_t p = 0; p < n_parts; ++p) {
            const lcsgpu_mst_key& k = keys[(size_t)p * n + v];
            if (less(k.dist_bits, k.id, bd, bi)) { bd = k.dist_bits; bi = k.id; }
        }
        if (bi == NO_ID) continue;
        const int32_t c = comp[v];
        if (c < 0 || c >= n) return fail(LCSGPU_E_INVALID, "component label %d of vertex %d out of range", c, v);
        if (less(bd, bi, cb_d[c], cb_id[c])) { cb_d[c] = bd; cb_id[c] = bi; }
    }
    std::vector<int32_t> parent((size_t)n);
    for (int32_t c = 0; c < n; ++c) parent[c] = c;
    int32_t found = *n_edges;
    for (int32_t c = 0; c < n; ++c) {
        if (comp[c] != c || cb_id[c] == NO_ID) continue;
        const uint64_t packed = ~cb_id[c];
        const int32_t x = (int32_t)(packed >> 32), y = (int32_t)(packed & 0xffffffffull);
        if (x < 0 || y < 0 || x >= n || y >= n) return fail(LCSGPU_E_INVALID, "key of component %d names the pair (%d, %d)", c, x, y);
        if ((comp[x] == c) == (comp[y] == c))
            return fail(LCSGPU_E_STATE, "MST: the key of component %d names the pair (%d, %d), which does not leave it "
                                        "(keys gathered from different rounds?)", c, x, y);
        const int32_t other = comp[x] == c ? comp[y] : comp[x];
        parent[c] = other;
        const bool mutual = cb_id[other] == cb_id[c];
        if (!mutual || c < other) {
            if (found >= n - 1) return fail(LCSGPU_E_STATE, "MST: more than n-1 edges");
            edges[found].from = x;
            edges[found].to = y;
            uint64_t bits = cb_d[c];
            memcpy(&edges[found].dist, &bits, 8);
            ++found;
        }
    }
    for (int32_t c = 0; c < n; ++c) { // two components that chose each other: the smaller one becomes the root
        if (comp[c] != c) continue;
        const int32_t p = parent[c];
        if (p != c && parent[p] == c && c < p) parent[c] = c;
    }
    for (int32_t v = 0; v < n; ++v) {
        int32_t r = comp[v], steps = 0;
        for (int32_t p = parent[r]; p != r; p = parent[r]) {
            r = p;
            if (++steps > n) return fail(LCSGPU_E_STATE, "MST: inconsistent keys (a hooking cycle among the components)");
        }
        cb_d[v] = (uint64_t)(uint32_t)r; // new labels, written back after every old one has been read
    }
    for (int32_t v = 0; v < n; ++v) comp[v] = (int32_t)cb_d[v];
    *n_edges = found;
    return LCSGPU_OK;
}

int lcsgpu_mst_order_edges(lcsgpu_mst_edge* edges, int32_t n)
{
    if (n >= 2 && !edges) return fail(LCSGPU_E_INVALID, "NULL edges");
    return order_edges_like_prim(edges, n);
}

int lcsgpu_mst_prim(lcsgpu_ctx* ctx, int distance_kind, lcsgpu_mst_edge* out_edges)
{
    const bool triangle_orientation = (distance_kind & LCSGPU_MST_TRIANGLE_ORIENTATION) != 0;
    distance_kind &= ~LCSGPU_MST_TRIANGLE_ORIENTATION;
    if (!ctx) return fail(LCSGPU_E_INVALID, "NULL ctx");
    if (ctx->n < 0) return fail(LCSGPU_E_STATE, "no sequence set uploaded");
    if (!valid_kind(distance_kind)) return fail(LCSGPU_E_INVALID, "unknown distance kind %d", distance_kind);
    const int32_t n = ctx->n;
    if (n < 2) return LCSGPU_OK;
    if (!out_edges) return fail(LCSGPU_E_INVALID, "NULL out_edges");
    LaneGuard guard(ctx, LaneGuard::LANE0);
    Lane& L = guard.lane();
    HIP_TRY(hipSetDevice(ctx->device));
    const int elem = ctx->max_len > 65535 ? 4 : 2;
    const size_t pairs = (size_t)n * (n - 1) / 2;
    int32_t nq = 0;
    for (int32_t i = 0; i < n && !triangle_orientation; ++i) nq += ctx->quirk[i] ? 1 : 0;
    int rc;

    const char* mode_env = getenv("LCSGPU_MST_MODE"); // passes | fused | recompute (below), prim = the step-by-step kernel on any set
    if ((triangle_orientation || nq == 0) && !(mode_env && !strcmp(mode_env, "prim"))) {
        // Distances do not depend on which endpoint is the ref: Boruvka rounds (one block = all rows, no exchange),
        // then Prim's insertion order as a walk over the n-1 tree edges.  Where the rounds get their LCS values from:
        //   passes    (default while the triangle fits) the triangle is computed into HBM, every round streams it
        //   recompute (when it does not fit -- n > ~530 000 on 288 GB -- or on request) no triangle: every round
        //             recomputes the LCS values with the fold fused into the launch; O(n) device memory
        //   fused     (on request) the triangle is computed by a launch that does round 0's local half as well.
        //             Measured at 100 000 x 400 aa: the fold's 5 extra registers leave the 13-half-word kernel's
        //             register pass no free seat (174 instead of 28 of its 1040 three-source ops keep a shared bank)
        //             -- LCS launch 1318 -> 1338 ms for 4 ms of passes saved -- so it is not the default.
        // LCSGPU_MST_MODE=passes|fused|recompute overrides the choice (tests, measurements).
        enum { AUTO, FUSED, RECOMPUTE, PASSES } mode = AUTO;
        if (const char* e = mode_env)
            mode = !strcmp(e, "fused") ? FUSED : !strcmp(e, "recompute") ? RECOMPUTE : !strcmp(e, "passes") ? PASSES : AUTO;
        if (elem != 2) mode = PASSES;
        // rows [0, r_fit) of the triangle are kept in HBM, rows [r_fit, n) are recomputed every round
        int32_t r_fit = mode == RECOMPUTE ? 0 : n;
        if (r_fit == n) {
            rc = reserve_big(ctx, L.d_out, pairs * elem, "the LCS triangle of the MST");
            if (rc == LCSGPU_E_NOMEM && mode == AUTO) {
                // No room for 2 B per pair.  Keep the rows that do fit -- every round then recomputes only the rest
                // (at 600 000 sequences on 288 GB: the last 80 000 rows, a quarter of the pairs) -- or, when that would
                // be less than a quarter of the rows, nothing at all.
                size_t free_b = 0;
                if (int rc2 = device_free_bytes(ctx, &free_b)) return rc2;
                free_b += L.d_out.cap;
                // what else must fit: the MST state (n x ~200 B and the column pass's partials, 96 chunks x n x 16 B), plans, slack
                const size_t keep_free = (size_t)n * (256 + 96 * 16) + std::min<size_t>((size_t)2 << 30, free_b / 16);
                const double usable = free_b > keep_free ? (double)(free_b - keep_free) : 0.0;
                r_fit = (int32_t)std::min<double>(n, std::floor(0.5 + std::sqrt(0.25 + 2.0 * usable / elem)));
                if (r_fit < n / 4) r_fit = 0;
                while (r_fit > 0 && reserve_big(ctx, L.d_out, (size_t)tri_offset(r_fit) * elem, "the resident rows of the LCS triangle"))
                    r_fit = r_fit * 15 / 16 < n / 4 ? 0 : r_fit * 15 / 16;
            } else if (rc)
                return rc;
        }
        if (r_fit == n && mode != FUSED) {
            rc = run_rows(ctx, L, lcsgpu::MODE_TRIANGLE, nullptr, 0, n, nullptr, 0, n - 1, L.d_out.p, 0, 0, elem);
            if (!rc) rc = shard_begin(ctx, L, L.d_out.p, elem, 0, n, distance_kind);
        } else if (r_fit == n || r_fit == 0)
            rc = shard_begin(ctx, L, r_fit ? L.d_out.p : nullptr, elem, 0, n, distance_kind, true);
        else {
            rc = run_rows(ctx, L, lcsgpu::MODE_TRIANGLE, nullptr, 0, r_fit, nullptr, 0, r_fit - 1, L.d_out.p, 0, 0, elem);
            if (!rc) rc = shard_begin(ctx, L, L.d_out.p, elem, 0, r_fit, distance_kind);
            if (!rc && ctx->d_gather.reserve(2 * (size_t)n * sizeof(lcsgpu_mst_key)) != hipSuccess)
                rc = fail(LCSGPU_E_NOMEM, "no device memory for the keys of the two row blocks");
        }
        const bool hybrid = r_fit > 0 && r_fit < n;
        if (getenv("LCSGPU_PROFILE"))
            fprintf(stderr, "lcsgpu_mst_prim: n = %d, rows [0, %d) of the triangle resident (%.2f GB), rows [%d, %d) recomputed per round%s\n",
                    n, r_fit, (double)tri_offset(r_fit) * elem / 1e9, r_fit, n,
                    r_fit == n ? (mode == FUSED ? " -- round 0 fused into the launch" : "") : r_fit == 0 ? " -- no triangle" : " -- hybrid");
        while (!rc && ctx->mst.found < n - 1) {
            if (!hybrid) {
                rc = shard_best(ctx, L, nullptr, nullptr);
                if (!rc) rc = shard_merge(ctx, L, nullptr, 1, nullptr);
                continue;
            }
            // two row blocks on one GPU: [0, r_fit) by passes over its resident triangle, [r_fit, n) by a launch with the
            // fold fused in; their keys meet like two ranks' keys (part 0, part 1 of the gathered buffer)
            lcsgpu::MstKey* gathered = (lcsgpu::MstKey*)ctx->d_gather.p;
            rc = shard_best(ctx, L, gathered, nullptr);
            if (rc) break;
            lcsgpu::BoruvkaArgs rest = ctx->mst.b; // the component state is shared; the block and where its keys go differ
            rest.tri = nullptr;
            rest.r0 = r_fit;
            rest.r1 = n;
            rest.off = tri_offset(r_fit);
            rest.best = gathered + n;
            rc = fused_rows(ctx, L, rest, nullptr, elem, ctx->mst.rounds > 0);
            if (rc) break;
            if (const hipError_t e = lcsgpu::launch_boruvka_fuse_fold(rest, L.stream)) {
                rc = fail(LCSGPU_E_HIP, "the fold of the recomputed rows failed: %s", hipGetErrorString(e));
                break;
            }
            rc = shard_merge(ctx, L, gathered, 2, nullptr);
        }
        const auto t_fin = std::chrono::steady_clock::now();
        if (!rc) rc = shard_finish(ctx, L, out_edges);
        if (getenv("LCSGPU_PROFILE") || getenv("LCSGPU_MST_COUNTS")) {
            fprintf(stderr, "lcsgpu_mst_prim: %d rounds; edges to the host + Prim's order %.3f s\n", ctx->mst.rounds,
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t_fin).count());
            unsigned long long st[2] = {0, 0};
            if (!rc && hipMemcpy(st, ctx->mst.b.counters + 8, 16, hipMemcpyDeviceToHost) == hipSuccess && st[0] + st[1] > 0)
                // (the reference's "No. comp. necessary / useless", tree/MSTPrim.cpp:544-546, by workgroup tiles of <= 256 x 32 pairs)
                fprintf(stderr, "lcsgpu_mst_prim: LCS tiles of the rounds that recompute: %llu computed, %llu let go by the length bound (%.1f %%)\n",
                        st[0], st[1], 100.0 * (double)st[1] / (double)(st[0] + st[1]));
        }
        ctx->mst.active = false; // the triangle it points to belongs to this call
        if (rc) return rc;
        note_async_call(ctx);
        return LCSGPU_OK;
    }

    // MSTPrim's own orientation with orientation-sensitive sequences in the set: the step-by-step kernel over the
    // resident triangle, the sensitive sequences' values in both roles as side tables
    rc = reserve_big(ctx, L.d_out, pairs * elem, "the LCS triangle of the MST");
    if (rc) return rc;
    rc = run_rows(ctx, L, lcsgpu::MODE_TRIANGLE, nullptr, 0, n, nullptr, 0, n - 1, L.d_out.p, 0, 0, elem);
    if (rc) return rc;
    std::vector<int32_t> qindex(n, -1), qlist;
    for (int32_t i = 0; i < n && !triangle_orientation; ++i)
        if (ctx->quirk[i]) {
            qindex[i] = (int32_t)qlist.size();
            qlist.push_back(i);
        }
    nq = (int32_t)qlist.size();
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

} // extern "C"

// ---- several contexts (one per GPU) working on ONE problem ------------------------------------------
namespace {

// lane 0 of every context, taken in address order (two multi-context calls cannot deadlock each other)
struct MultiGuard {
    std::vector<std::unique_ptr<LaneGuard>> guards; // index = position in the caller's context list
    std::vector<Lane*> lanes;
    MultiGuard(lcsgpu_ctx* const* ctxs, int n)
    {
        std::vector<int> order(n);
        for (int i = 0; i < n; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return ctxs[a] < ctxs[b]; });
        guards.resize(n);
        lanes.resize(n);
        for (int i : order) {
            guards[i].reset(new LaneGuard(ctxs[i], LaneGuard::LANE0));
            lanes[i] = &guards[i]->lane();
        }
    }
};

int check_multi(lcsgpu_ctx* const* ctxs, int32_t n_ctx)
{
    if (!ctxs || n_ctx < 1) return fail(LCSGPU_E_INVALID, "no contexts");
    for (int k = 0; k < n_ctx; ++k) {
        if (!ctxs[k]) return fail(LCSGPU_E_INVALID, "NULL ctx");
        if (ctxs[k]->n < 0) return fail(LCSGPU_E_STATE, "no sequence set uploaded");
        for (int j = 0; j < k; ++j)
            if (ctxs[j] == ctxs[k]) return fail(LCSGPU_E_INVALID, "the same context twice");
        if (ctxs[k]->n != ctxs[0]->n || ctxs[k]->lens != ctxs[0]->lens)
            return fail(LCSGPU_E_STATE, "context %d holds a different sequence set than context 0", k);
    }
    return LCSGPU_OK;
}

// row-block boundaries with (nearly) equal pair counts over rows [r0, r1): row i holds i pairs
std::vector<int32_t> equal_pair_cuts(int32_t r0, int32_t r1, int parts)
{
    std::vector<int32_t> cut(parts + 1, r1);
    cut[0] = r0;
    const double p0 = (double)r0 * (r0 - 1) / 2, p1 = (double)r1 * (r1 - 1) / 2;
    for (int k = 1; k < parts; ++k) {
        const double target = p0 + (p1 - p0) * k / parts; // pairs below the cut
        const int32_t r = (int32_t)std::floor(0.5 + std::sqrt(0.25 + 2.0 * std::max(0.0, target)));
        cut[k] = std::min(r1, std::max(cut[k - 1], r));
    }
    return cut;
}

// ---- device-to-device transport between the contexts of one process ---------------------------------------
// Peer access is a property of an ordered device pair and has to be switched on once per process; without it
// hipMemcpyPeerAsync still works but is staged through host memory by the runtime, i.e. PCIe both ways instead
// of one xGMI hop (7 links x ~153 GB/s per GPU).  When the devices cannot address each other at all the copy goes
// through a pinned host buffer here, in 64 MB pieces (two PCIe crossings, ~25 GB/s: the 8.75 GB a GPU receives for
// lcsgpu_multi_upgma at 100 000 sequences then take ~0.35 s instead of ~0.06 s).
//   LCSGPU_TRANSPORT=peer   same-device contexts take the hipMemcpyPeerAsync branch too (so a 1-GPU box runs it)
//   LCSGPU_TRANSPORT=host   every copy between contexts takes the pinned-host path (the no-peer-access fallback)
std::mutex g_peer_mu;
std::vector<std::pair<std::pair<int, int>, bool>> g_peer; // (accessing device, owner of the memory) -> usable

bool transport_forced(const char* which)
{
    const char* e = getenv("LCSGPU_TRANSPORT");
    return e && !strcmp(e, which);
}

// may kernels / copy engines of device `from` address memory of device `to`?  Switches it on at first use.
bool peer_access(int from, int to)
{
    if (from == to) return true;
    std::lock_guard<std::mutex> lk(g_peer_mu);
    for (const auto& e : g_peer)
        if (e.first.first == from && e.first.second == to) return e.second;
    int can = 0;
    bool ok = hipDeviceCanAccessPeer(&can, from, to) == hipSuccess && can;
    if (ok) {
        int cur = 0;
        (void)hipGetDevice(&cur);
        ok = hipSetDevice(from) == hipSuccess;
        if (ok) {
            const hipError_t e = hipDeviceEnablePeerAccess(to, 0);
            ok = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
        }
        (void)hipGetLastError();
        (void)hipSetDevice(cur);
    } else
        (void)hipGetLastError();
    g_peer.push_back({{from, to}, ok});
    if (getenv("LCSGPU_PROFILE")) // once per ordered pair and process
        fprintf(stderr, "lcsgpu: device %d -> device %d: %s\n", from, to,
                ok ? "peer access on (copies are hipMemcpyPeerAsync, one xGMI hop)" : "no peer access (copies are staged through pinned host memory)");
    return ok;
}

// `bytes` from src (device memory of src_ctx) to dst (device memory of dst_ctx), ordered on `stream` (a stream of
// src_ctx's device: the producer pushes).  Returns with the copy queued -- or, on the staging path, done.
int copy_between(lcsgpu_ctx* dst_ctx, void* dst, lcsgpu_ctx* src_ctx, const void* src, size_t bytes, hipStream_t stream)
{
    if (!bytes) return LCSGPU_OK;
    const int sd = src_ctx->device, dd = dst_ctx->device;
    HIP_TRY(hipSetDevice(sd));
    if (transport_forced("host") || (sd != dd && !(peer_access(sd, dd) && peer_access(dd, sd)))) {
        static std::mutex mu; // one staging buffer per process: this path is the exception, not the design
        static PinBuf stage;
        std::lock_guard<std::mutex> lk(mu);
        const size_t piece = (size_t)64 << 20;
        HIP_TRY(stage.reserve(std::min(bytes, piece)));
        for (size_t at = 0; at < bytes; at += piece) {
            const size_t m = std::min(piece, bytes - at);
            HIP_TRY(hipSetDevice(sd));
            HIP_TRY(hipMemcpyAsync(stage.p, (const char*)src + at, m, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            HIP_TRY(hipSetDevice(dd));
            HIP_TRY(hipMemcpy((char*)dst + at, stage.p, m, hipMemcpyHostToDevice));
        }
        HIP_TRY(hipSetDevice(sd));
        return LCSGPU_OK;
    }
    if (sd == dd && !transport_forced("peer"))
        HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream));
    else
        HIP_TRY(hipMemcpyPeerAsync(dst, dd, src, sd, bytes, stream));
    return LCSGPU_OK;
}

// ---- the key exchange of the Boruvka rounds as an RCCL all-gather ---------------------------------------------------
// One communicator per context of the call (ncclCommInitAll: one process, one rank per device), cached by device list
// for the life of the process; librccl (0.5 GB, seconds to initialise) is loaded only when this is asked for, so the
// library keeps depending on the HIP runtime alone.  A round's exchange is then ONE grouped ncclAllGather, in place
// (rank k's keys already sit in slot k of its own gather buffer), on the lanes' streams: no events, no copies by hand.
// RCCL wants one device per rank: contexts that share a device cannot use it (the peer-copy form handles those).
//
// Which exchange lcsgpu_multi_mst_prim uses (LCSGPU_EXCHANGE = auto | rccl | peer; unset = auto):
//   auto  the grouped ncclAllGather when there are >= 2 contexts, every context sits on its own device, librccl loads and
//         the communicators initialise -- the north star's "RCCL allgather of per-row minima over xGMI"; else the peer
//         copies, and g_exchange_note says why (lcsgpu_multi_transport reports it).
//   rccl  the all-gather or an error (also with ONE context, so that a 1-GPU box executes the RCCL calls)
//   peer  the N x (N-1) peer copies of copy_between
// The first round a communicator set ever serves is cross-checked: the same round's keys also travel by peer copies into
// the idle half of the gathered buffers and every context's two halves are compared on the host, byte for byte; a
// difference ends the call with LCSGPU_E_HIP naming the context and the slot (LCSGPU_EXCHANGE_CHECK=0 skips the check,
// =always repeats it every call).
struct RcclComms {
    std::vector<ncclComm_t> comm;
    bool verified = false; // the cross-check against the peer copies has passed once
};
struct Rccl {
    void* lib = nullptr;
#if LCSGPU_HAVE_RCCL_HEADERS
    decltype(&ncclCommInitAll) comm_init_all = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclGroupStart) group_start = nullptr;
    decltype(&ncclGroupEnd) group_end = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
#endif
    std::map<std::vector<int>, RcclComms> comms;
    std::string why; // why it is unusable
};
std::mutex g_rccl_mu;
Rccl g_rccl;
std::string g_exchange_note = "no multi-context single-linkage call yet"; // what the last lcsgpu_multi_mst_prim used, and why

enum class Exchange { AUTO, RCCL, PEER };
Exchange exchange_wanted()
{
    const char* e = getenv("LCSGPU_EXCHANGE");
    if (e && !strcmp(e, "rccl")) return Exchange::RCCL;
    if (e && !strcmp(e, "peer")) return Exchange::PEER;
    return Exchange::AUTO;
}

// communicators for these devices, in this order
int rccl_comms(const std::vector<int>& devices, RcclComms** out)
{
#if !LCSGPU_HAVE_RCCL_HEADERS
    (void)devices;
    (void)out;
    return fail(LCSGPU_E_UNSUPPORTED, "the RCCL exchange was compiled out: <rccl/rccl.h> was not found when liblcsgpu was built");
#else
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    Rccl& R = g_rccl;
    if (!R.lib && R.why.empty()) {
        for (const char* name : {"librccl.so.1", "librccl.so"})
            if ((R.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL))) break;
        if (!R.lib) R.why = std::string("dlopen(librccl.so.1): ") + dlerror();
        else {
            R.comm_init_all = (decltype(R.comm_init_all))dlsym(R.lib, "ncclCommInitAll");
            R.all_gather = (decltype(R.all_gather))dlsym(R.lib, "ncclAllGather");
            R.group_start = (decltype(R.group_start))dlsym(R.lib, "ncclGroupStart");
            R.group_end = (decltype(R.group_end))dlsym(R.lib, "ncclGroupEnd");
            R.error_string = (decltype(R.error_string))dlsym(R.lib, "ncclGetErrorString");
            if (!R.comm_init_all || !R.all_gather || !R.group_start || !R.group_end || !R.error_string) {
                R.why = "librccl lacks ncclCommInitAll / ncclAllGather / ncclGroupStart / ncclGroupEnd";
                R.lib = nullptr;
            }
        }
    }
    if (!R.lib) return fail(LCSGPU_E_UNSUPPORTED, "RCCL exchange: %s", R.why.c_str());
    for (size_t a = 0; a < devices.size(); ++a)
        for (size_t b = 0; b < a; ++b)
            if (devices[a] == devices[b])
                return fail(LCSGPU_E_UNSUPPORTED, "RCCL exchange: contexts %zu and %zu share device %d (RCCL wants one device per "
                                                  "rank; LCSGPU_EXCHANGE=peer or unset for the peer-copy exchange)", b, a, devices[a]);
    auto it = R.comms.find(devices);
    if (it == R.comms.end()) {
        RcclComms c;
        c.comm.assign(devices.size(), nullptr);
        const ncclResult_t e = R.comm_init_all(c.comm.data(), (int)devices.size(), devices.data());
        if (e != ncclSuccess) return fail(LCSGPU_E_HIP, "ncclCommInitAll over %zu devices failed: %s", devices.size(), R.error_string(e));
        it = R.comms.emplace(devices, std::move(c)).first;
    }
    *out = &it->second;
    return LCSGPU_OK;
#endif
}

// every context's keys (slot k of its own buffer) into every context's buffer: base[k] = the N-slot buffer of context k
int rccl_all_gather_keys(const std::vector<ncclComm_t>& comms, const std::vector<char*>& base, size_t key_bytes,
                         const std::vector<hipStream_t>& streams, const std::vector<int>& devices)
{
#if !LCSGPU_HAVE_RCCL_HEADERS
    (void)comms; (void)base; (void)key_bytes; (void)streams; (void)devices;
    return fail(LCSGPU_E_UNSUPPORTED, "the RCCL exchange was compiled out");
#else
    Rccl& R = g_rccl;
    ncclResult_t e = R.group_start();
    hipError_t he = hipSuccess;
    for (size_t k = 0; k < comms.size() && e == ncclSuccess && he == hipSuccess; ++k) {
        he = hipSetDevice(devices[k]);
        if (he == hipSuccess) e = R.all_gather(base[k] + k * key_bytes, base[k], key_bytes / 8, ncclUint64, comms[k], streams[k]);
    }
    const ncclResult_t e2 = R.group_end(); // whatever happened inside: the group must be closed
    if (e == ncclSuccess) e = e2;
    if (he != hipSuccess) return fail(LCSGPU_E_HIP, "hipSetDevice failed: %s", hipGetErrorString(he));
    if (e != ncclSuccess) return fail(LCSGPU_E_HIP, "ncclAllGather of the best-edge keys failed: %s", R.error_string(e));
    return LCSGPU_OK;
#endif
}

// how a copy between two contexts travels (copy_between's decision, without making the copy)
const char* transport_between(int sd, int dd)
{
    if (transport_forced("host")) return "host-staging(forced)";
    if (sd == dd) return transport_forced("peer") ? "peer-copy(same device, forced)" : "same-device";
    return peer_access(sd, dd) && peer_access(dd, sd) ? "peer-copy" : "host-staging";
}

// The whole LCS triangle of the uploaded set into lane 0's result buffer of ctxs[0]: row blocks of equal
// pair counts, one per context, each computed on its own GPU at the same time; the blocks of the other
// contexts travel into place over xGMI (copy_between: hipMemcpyPeerAsync on the producer's stream, peer access
// switched on for the pair) -- the "all-gather of
// u16 row blocks" of a matrix consumer that lives on one device.  No host thread per GPU is
// needed: every launch and copy is asynchronous.
int whole_triangle(lcsgpu_ctx* const* ctxs, int32_t n_ctx, MultiGuard& g, int elem)
{
    lcsgpu_ctx* c0 = ctxs[0];
    Lane& L0 = *g.lanes[0];
    const int32_t n = c0->n;
    int rc = reserve_big(c0, L0.d_out, (size_t)tri_offset(n) * elem, "the LCS triangle");
    if (rc) return rc;
    const std::vector<int32_t> cut = equal_pair_cuts(0, n, n_ctx);
    std::vector<hipEvent_t> landed(n_ctx, nullptr);
    struct Events {
        std::vector<hipEvent_t>& v;
        ~Events() { for (hipEvent_t e : v) if (e) (void)hipEventDestroy(e); }
    } events{landed};
    for (int k = 0; k < n_ctx; ++k) {
        const int32_t r0 = cut[k], r1 = cut[k + 1];
        if (r1 <= r0) continue;
        Lane& L = *g.lanes[k];
        void* dst = L0.d_out.p;
        int64_t off = 0;
        if (k > 0) {
            off = tri_offset(r0);
            rc = reserve_big(ctxs[k], L.d_out, (size_t)(tri_offset(r1) - off) * elem, "a row block of the LCS triangle");
            if (rc) return rc;
            dst = L.d_out.p;
        }
        rc = run_rows(ctxs[k], L, lcsgpu::MODE_TRIANGLE, nullptr, r0, r1 - r0, nullptr, 0, std::max(0, r1 - 1), dst, 0, off, elem, r0);
        if (rc) return rc;
        if (k > 0) {
            const size_t bytes = (size_t)(tri_offset(r1) - off) * elem;
            char* place = (char*)L0.d_out.p + (size_t)off * elem;
            rc = copy_between(c0, place, ctxs[k], L.d_out.p, bytes, L.stream); // xGMI with peer access (see copy_between)
            if (rc) return rc;
            HIP_TRY(hipSetDevice(ctxs[k]->device));
            HIP_TRY(hipEventCreateWithFlags(&landed[k], hipEventDisableTiming));
            HIP_TRY(hipEventRecord(landed[k], L.stream));
        }
    }
    HIP_TRY(hipSetDevice(c0->device));
    for (int k = 1; k < n_ctx; ++k)
        if (landed[k]) HIP_TRY(hipStreamWaitEvent(L0.stream, landed[k], 0));
    for (int k = 1; k < n_ctx; ++k) { // the staging buffers of the plans and the events must outlive the work
        if (!landed[k]) continue;
        HIP_TRY(hipEventSynchronize(landed[k]));
        g.lanes[k]->plan_in_flight = false;
    }
    return LCSGPU_OK;
}

// The merges of a reducer back on the host: flags | left | right lie next to each other on the device, so ONE copy into the
// lane's pinned buffer brings them (three copies into pageable memory took 8 ms after NJ's 66-ms launch: each is staged
// and waited for on its own).
int fetch_merges(Lane& L, const char* d_flags, size_t flags_bytes, size_t left_at, size_t right_at, int32_t n, void* flags,
                 int32_t* out_left, int32_t* out_right)
{
    const size_t total = right_at + (size_t)(n - 1) * 4;
    HIP_TRY(L.h_small.reserve(total));
    HIP_TRY(hipMemcpyAsync(L.h_small.p, d_flags, total, hipMemcpyDeviceToHost, L.stream));
    HIP_TRY(hipStreamSynchronize(L.stream));
    const char* h = (const char*)L.h_small.p;
    memcpy(flags, h, flags_bytes);
    memcpy(out_left, h + left_at, (size_t)(n - 1) * 4);
    memcpy(out_right, h + right_at, (size_t)(n - 1) * 4);
    return LCSGPU_OK;
}

// resident: the whole LCS triangle sits in L.d_out already (the multi-context gather); else this function computes it
// itself, in row blocks that are turned into float distances one after the other -- 2 B per pair never exist all at once.
int upgma_reduce(lcsgpu_ctx* ctx, Lane& L, int elem, int distance_kind, int modified, int32_t* out_left, int32_t* out_right, bool resident)
{
    const int32_t n = ctx->n;
    HIP_TRY(hipSetDevice(ctx->device));
    // The distances as a full symmetric matrix where that fits: both rows a merge reads are then contiguous
    // (tree_kernels.hip).  Else the packed triangle.  LCSGPU_UPGMA_LAYOUT=triangle|square forces one (tests, measurements).
    const auto t_entry = std::chrono::steady_clock::now();
    auto since_entry = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_entry).count(); };
    bool square = true;
    if (const char* e = getenv("LCSGPU_UPGMA_LAYOUT")) square = !strcmp(e, "square");
    // Several merges per launch (upgma_batch_kernels.hip) -- the default while its layout fits: n rows x (n + spare) SLOTS (a
    // new cluster keeps its left child's row but gets a new column, so that a batch's columns are consecutive; when the spare
    // is used up the live slots are compacted: 44 GB at 100 000 sequences, ~255 000 sequences on 288 GB).  A batch = the
    // next <= K entries of the rows' sorted (min_dist, index) order, computed together and committed as far as the reference
    // would have picked them in that order (practically always all K).  LCSGPU_UPGMA_BATCH=0 keeps one launch per merge on
    // the n x n matrix; = 8 | 16 | 32 selects K (32).  LCSGPU_TUNE upgma_spare=<slots>: the spare (tests: a small one
    // compacts every few batches).
    int batch_k = 32;
    if (const char* e = getenv("LCSGPU_UPGMA_BATCH")) batch_k = atoi(e);
    batch_k = batch_k >= 32 ? 32 : batch_k >= 16 ? 16 : batch_k >= 8 ? 8 : 0;
    if (n < 3) batch_k = 0;
    // the LCS values of a row block (not resident): at most 2^30 pairs at a time
    const int64_t block_pairs = std::min<int64_t>(tri_offset(n), (int64_t)1 << 30);
    const size_t block_bytes = resident ? 0 : (size_t)block_pairs * elem;
    size_t avail = 0;
    {
        int rc0 = device_free_bytes(ctx, &avail);
        if (rc0) return rc0;
        avail += ctx->d_dist.cap + (resident ? 0 : L.d_out.cap); // what these two buffers hold already is theirs to use
        avail -= std::min(avail, block_bytes + ((size_t)64 << 20)); // the block's values and the small arrays come first
    }
    size_t ld = (size_t)n;
    int rc = LCSGPU_E_NOMEM;
    // (Round 6 tried to hide the matrix's allocation: 44 GB are handed out at once on a device that has rested and in 1.2-2.1 s when
    //  another process has just given that much back.  A helper thread allocating while this one computed the whole uint16
    //  triangle hid nothing -- the launches of this process stand still while the driver hands the memory out: 1.21 s of
    //  allocation + 1.37 s of LCS = 2.84 s to the first distance launch, as before -- and cost 10 GB more;
    //  profiles/upgma_modes_r06.txt.)
    if (square && batch_k) {
        const int forced = tune_int("upgma_spare", 0);
        for (size_t div : {10, 20, 40}) {
            size_t spare = forced > 0 ? (size_t)forced : std::max<size_t>(2048, (size_t)n / div);
            spare = std::max<size_t>(spare, (size_t)2 * batch_k);
            ld = ((size_t)n + spare + 63) & ~(size_t)63;
            if ((size_t)n * ld * sizeof(float) <= avail) {
                rc = reserve_big(ctx, ctx->d_dist, (size_t)n * ld * sizeof(float), "the float distance matrix (rows x slots)");
                if (rc != LCSGPU_E_NOMEM) break;
            }
            if (forced > 0) break;
        }
        if (rc == LCSGPU_E_NOMEM) {
            batch_k = 0;
            ld = (size_t)n;
        }
    }
    if (square && !batch_k && (size_t)n * n * sizeof(float) <= avail)
        rc = reserve_big(ctx, ctx->d_dist, (size_t)n * n * sizeof(float), "the float distance matrix");
    if (rc == LCSGPU_E_NOMEM) {
        batch_k = 0;
        square = false;
        ld = (size_t)n;
        rc = reserve_big(ctx, ctx->d_dist, (size_t)tri_offset(n) * sizeof(float), "the float distance triangle");
    }
    if (rc) return rc;
    if (!resident) {
        rc = reserve_big(ctx, L.d_out, block_bytes, "a row block of the LCS triangle");
        if (rc) return rc;
    }
    const int blocks = (n + 255) / 256;
    auto a16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t o_min = 0, o_near = o_min + a16((size_t)n * 4), o_node = o_near + a16((size_t)n * 4),
                 o_pd = o_node + a16((size_t)n * 4), o_pj = o_pd + a16((size_t)2 * blocks * 4),
                 o_bd = o_pj + a16((size_t)2 * blocks * 4), o_bj = o_bd + a16((size_t)blocks * 4),
                 o_bn = o_bj + a16((size_t)blocks * 4), o_sel = o_bn + a16((size_t)blocks * 4), o_left = o_sel + 256,
                 o_right = o_left + a16((size_t)n * 4), total = o_right + a16((size_t)n * 4);
    HIP_TRY(ctx->d_prim.reserve(total));
    char* base = (char*)ctx->d_prim.p;
    HIP_TRY(hipMemsetAsync(base + o_sel, 0, o_left - o_sel, L.stream)); // flags
    lcsgpu::UpgmaArgs a{};
    a.bm_d = (float*)(base + o_bd);
    a.bm_j = (uint32_t*)(base + o_bj);
    a.bm_near = (uint32_t*)(base + o_bn);
    a.D = (float*)ctx->d_dist.p;
    a.square = square ? 1 : 0;
    a.ld = (int64_t)ld;
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
    const bool profile = getenv("LCSGPU_PROFILE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_pro = 0, t_merge = 0;
    if (profile) {
        HIP_TRY(hipStreamSynchronize(L.stream));
        t_pro = now();
        fprintf(stderr, "lcsgpu_upgma: %.3f s from the call to the first launch of the distances (the buffers' allocation)\n", since_entry());
    }
    // the LCS values -> float distances, row block by row block (whole tile rows of 32 in the square layout)
    std::vector<hipEvent_t> t_ev;
    struct EvGuard {
        std::vector<hipEvent_t>& v;
        ~EvGuard() { for (hipEvent_t e : v) if (e) (void)hipEventDestroy(e); }
    } ev_guard{t_ev};
    int lcs_launches = 0;
    if (resident) {
        HIP_TRY(lcsgpu::launch_upgma_distances(a, L.d_out.p, elem, (const uint32_t*)ctx->d_lens.p, (const float*)ctx->d_powf.p,
                                               distance_kind, 0, n, L.stream));
    } else {
        for (int32_t r0 = 0; r0 < n;) {
            int32_t r1 = n;
            if (tri_offset(n) - tri_offset(r0) > block_pairs) { // the largest multiple of 32 whose rows fit the block
                r1 = (int32_t)std::floor(0.5 + std::sqrt(0.25 + 2.0 * (double)(tri_offset(r0) + block_pairs)));
                r1 = std::min(n, r1) & ~31;
                while (r1 > r0 + 32 && tri_offset(r1) - tri_offset(r0) > block_pairs) r1 -= 32;
                if (r1 <= r0) r1 = std::min(n, r0 + 32);
                if (tri_offset(r1) - tri_offset(r0) > block_pairs)
                    return fail(LCSGPU_E_NOMEM, "UPGMA: a block of 32 rows of the LCS triangle does not fit its buffer (n = %d)", n);
            }
            hipEvent_t e0 = nullptr, e1 = nullptr;
            HIP_TRY(hipEventCreate(&e0));
            t_ev.push_back(e0);
            HIP_TRY(hipEventCreate(&e1));
            t_ev.push_back(e1);
            HIP_TRY(hipEventRecord(e0, L.stream));
            rc = run_rows(ctx, L, lcsgpu::MODE_TRIANGLE, nullptr, r0, r1 - r0, nullptr, 0, std::max(0, r1 - 1), L.d_out.p, 0, tri_offset(r0),
                          elem, r0);
            if (rc) return rc;
            lcs_launches += L.last_launches;
            HIP_TRY(hipEventRecord(e1, L.stream));
            HIP_TRY(lcsgpu::launch_upgma_distances(a, L.d_out.p, elem, (const uint32_t*)ctx->d_lens.p, (const float*)ctx->d_powf.p,
                                                   distance_kind, r0, r1, L.stream));
            r0 = r1; // (run_rows waits for the stream before it reuses the lane's plan staging)
        }
    }
    HIP_TRY(lcsgpu::launch_upgma_init(a, L.stream));
    if (profile) { HIP_TRY(hipStreamSynchronize(L.stream)); t_merge = now(); }
    // The n-1 merges: batches of up to K merges per three launches (upgma_batch_kernels.hip), else one launch per merge.
    // (Round 3 also had all merges inside one kernel on one XCD: bit-identical, 11-13 us per merge at 100 000 sequences
    // against 8.4 us for a launch -- profiles/upgma_chain_ab_r03.txt; removed in round 5.)
    uint32_t sel[12] = {0};
    bool merged = false;
    int n_batches = 0, n_cut = 0, n_compactions = 0;
    if (!merged && square && batch_k) {
        const size_t nb = (ld + 255) / 256;
        const size_t b_s0 = 0, b_s1 = b_s0 + a16(((size_t)n + 1) * 8), b_pos = b_s1 + a16(((size_t)n + 1) * 8),
                     b_slot = b_pos + a16((size_t)n * 4), b_rowof = b_slot + a16((size_t)n * 4),
                     b_cand = b_rowof + a16(ld * 4), b_state = b_cand + a16((size_t)lcsgpu::UPGMA_BATCH_CAND * 16),
                     b_hdr = b_state + 256, b_rec = b_hdr + 2048, b_side = b_rec + 8192, b_pd = b_side + a16((size_t)lcsgpu::UPGMA_BATCH_MAX * ld * 4),
                     b_pj = b_pd + a16((size_t)lcsgpu::UPGMA_BATCH_MAX * nb * 4), b_remap = b_pj + a16((size_t)lcsgpu::UPGMA_BATCH_MAX * nb * 4),
                     b_total = b_remap + a16(ld * 4);
        HIP_TRY(ctx->d_qrows.reserve(b_total)); // (a buffer the UPGMA path does not otherwise use)
        char* bb = (char*)ctx->d_qrows.p;
        lcsgpu::UpgmaBatchArgs ba{};
        ba.D = a.D;
        ba.ld = (int64_t)ld;
        ba.slot_of = (uint32_t*)(bb + b_slot);
        ba.row_of = (uint32_t*)(bb + b_rowof);
        ba.min_dist = a.min_dist;
        ba.nearest = a.nearest;
        ba.node_index = a.node_index;
        ba.left = a.left;
        ba.right = a.right;
        ba.n = n;
        ba.n_blocks = (int32_t)nb;
        ba.sorted0 = (uint2*)(bb + b_s0);
        ba.sorted1 = (uint2*)(bb + b_s1);
        ba.pos = (uint32_t*)(bb + b_pos);
        ba.cand = (uint4*)(bb + b_cand);
        ba.state = (uint32_t*)(bb + b_state);
        ba.hdr = (uint32_t*)(bb + b_hdr);
        ba.rec = (uint32_t*)(bb + b_rec);
        ba.side = (float*)(bb + b_side);
        ba.part_d = (float*)(bb + b_pd);
        ba.part_j = (uint32_t*)(bb + b_pj);
        ba.remap = (uint32_t*)(bb + b_remap);
        HIP_TRY(hipMemsetAsync(bb + b_state, 0, 256 + 2048, L.stream));
        HIP_TRY(lcsgpu::launch_upgma_batch_init(ba, L.stream));
        // The host does not know how many batches it takes (the validity check may cut one short), and the slots run out:
        // enqueue what the remaining merges need if every batch is full -- as far as the free slots reach --, look at the
        // committed count and the next free slot, compact when fewer than a batch's worth of slots are left, repeat.
        // Batches past the end do nothing.
        uint32_t st[8] = {0};
        int done = 0;
        long long slots_used = n; // the next free slot (exact after every look)
        while (done < n - 1) {
            if ((long long)ld - slots_used < batch_k) {
                HIP_TRY(lcsgpu::launch_upgma_compact(ba, n_batches & 1, slots_used, L.stream));
                slots_used = n - done; // = the live clusters
                ++n_compactions;
            }
            const int fit = (int)(((long long)ld - slots_used) / batch_k);
            const int want = std::min(fit, (n - 1 - done + batch_k - 1) / batch_k + (n_batches ? 2 : 0));
            HIP_TRY(lcsgpu::launch_upgma_batches(ba, modified != 0, batch_k, n_batches, want, slots_used, L.stream));
            n_batches += want;
            HIP_TRY(hipMemcpyAsync(st, ba.state + 8 * (n_batches & 1), 32, hipMemcpyDeviceToHost, L.stream));
            HIP_TRY(hipStreamSynchronize(L.stream));
            if (resident && L.d_out.cap >= ((size_t)1 << 30)) L.d_out.release(); // the gathered triangle has been consumed
            if (st[2]) {
                sel[8] = 1;
                break;
            }
            if ((int)st[0] <= done && want > 0 && (int)st[0] < n - 1)
                return fail(LCSGPU_E_STATE, "UPGMA: %d batches committed nothing (%u of %d merges)", want, st[0], n - 1);
            done = (int)st[0];
            n_cut = (int)st[3];
            slots_used = (long long)st[4];
            if (slots_used > (long long)ld || slots_used < (long long)n - done)
                return fail(LCSGPU_E_STATE, "UPGMA: slot bookkeeping out of range (%lld of %zu slots, %d merges)", slots_used, ld, done);
        }
        merged = true;
        if (st[2]) {
            L.plan_in_flight = false;
            return fail(LCSGPU_E_INVALID, "UPGMA: no finite nearest neighbour (a pair with LCS 0?) -- the reference's "
                                          "algorithm is undefined for this input");
        }
    }
    const bool batched = n_batches > 0;
    if (!merged) HIP_TRY(lcsgpu::launch_upgma_steps(a, modified != 0, L.stream));
    rc = fetch_merges(L, base + o_sel, 48, o_left - o_sel, o_right - o_sel, n, sel, out_left, out_right);
    if (rc) return rc;
    L.plan_in_flight = false;
    if (profile) fprintf(stderr, "lcsgpu_upgma: %.3f s from the call to the merges on the host\n", since_entry());
    if (profile)
        fprintf(stderr, "lcsgpu_upgma: n = %d, %s layout, distances + row minima %.3f s, %d merges %.3f s = %.2f us each (%s)\n", n,
                square ? "square" : "triangle", t_merge - t_pro, n - 1, now() - t_merge, 1e6 * (now() - t_merge) / std::max(n - 1, 1),
                batched ? "batches of merges, three launches each" : "one launch per merge");
    if (profile && batched)
        fprintf(stderr, "lcsgpu_upgma: %d batches of <= %d merges (%.1f merges per batch, %d cut short by a new row's key), %zu slots for %d rows, %d compactions\n",
                n_batches, batch_k, (double)(n - 1) / n_batches, n_cut, ld, n, n_compactions);
    if (sel[8])
        return fail(LCSGPU_E_INVALID, "UPGMA: no finite nearest neighbour (a pair with LCS 0?) -- the reference's "
                                      "algorithm is undefined for this input");
    if (resident) {
        note_async_call(ctx);
    } else { // this call's LCS launches, block by block (the stream has been synchronised)
        double ms = 0;
        for (size_t k = 0; k + 1 < t_ev.size(); k += 2) {
            float f = 0.f;
            if (hipEventElapsedTime(&f, t_ev[k], t_ev[k + 1]) == hipSuccess) ms += f;
        }
        g_last.ctx = ctx;
        g_last.also.clear();
        g_last.pending_on_lane0 = false;
        g_last.ms = ms;
        g_last.launches = lcs_launches;
        std::lock_guard<std::mutex> lk(ctx->mu);
        ctx->total_kernel_ms += ms;
    }
    return LCSGPU_OK;
}

// One resident launch at a time in this process: each wants every CU's LDS, and two of them -- two host threads with a context
// each on one device -- would hold half the chip each and wait for the other half.  (Another PROCESS on the same device can
// still do that: the workgroups give up after ~1 s and the call below runs the merges as launches.)
static std::mutex g_resident_launch;

int nj_reduce(lcsgpu_ctx* ctx, Lane& L, int elem, int distance_kind, int32_t* out_left, int32_t* out_right, bool resident = true)
{
    const int32_t n = ctx->n;
    HIP_TRY(hipSetDevice(ctx->device));
    const bool profile = getenv("LCSGPU_PROFILE") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_lap[5] = {0, 0, 0, 0, 0};
    if (profile) { HIP_TRY(hipStreamSynchronize(L.stream)); t_lap[0] = now(); }
    // One resident launch for all the merges (nj_loop_kernels.hip) while the rows' sums fit a workgroup's LDS and the
    // kernel fits a CU; else four launches per merge (tree_kernels.hip).  The same floats either way.
    const int cap = lcsgpu::nj_loop_cap(n);
    int grid = 0;
    if (resident && n >= 3 && n <= lcsgpu::NJ_LOOP_MAX_N && tune_int("nj_loop", 1)) {
        HIP_TRY(lcsgpu::nj_loop_grid(cap, &grid));
        const int g = tune_int("nj_groups", 0);
        if (g > 0 && g < grid) grid = g;
        if (grid > 0 && tune_int("nj_oversubscribe", 0)) grid *= 2; // a test aid: half the workgroups cannot be resident -- the others give up, the merges run as launches
    }
    const size_t tri_floats = ((size_t)tri_offset(n) + 3 + 4) & ~(size_t)3; // whole 16-byte loads at the end
    int rc = reserve_big(ctx, ctx->d_dist, tri_floats * sizeof(float) * (grid ? 2 : 1), "the float distance triangle");
    if (rc == LCSGPU_E_NOMEM && grid) { // no room for the second triangle the resident launch squeezes into: the launches need one
        grid = 0;
        rc = reserve_big(ctx, ctx->d_dist, tri_floats * sizeof(float), "the float distance triangle");
    }
    if (rc) return rc;
    auto a16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t o_sum = 0, o_tmp = o_sum + a16((size_t)n * 4), o_pq = o_tmp + a16((size_t)n * 4),
                 o_pi = o_pq + a16((size_t)n * 4), o_node = o_pi + a16((size_t)n * 4), o_act = o_node + a16((size_t)n * 4),
                 o_sel = o_act + a16((size_t)n), o_left = o_sel + 16, o_right = o_left + a16((size_t)n * 4),
                 o_slots = o_right + a16((size_t)n * 4), o_u = o_slots + (size_t)grid * 32 + 64,
                 total = o_u + (grid ? a16((size_t)n * 8) : 0);
    HIP_TRY(ctx->d_prim.reserve(total));
    char* base = (char*)ctx->d_prim.p;
    HIP_TRY(hipMemsetAsync(base + o_sel, 0, 16, L.stream));
    if (grid) HIP_TRY(hipMemsetAsync(base + o_slots, 0, total - o_slots, L.stream));
    if (profile) t_lap[1] = now();
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
    if (profile) { HIP_TRY(hipStreamSynchronize(L.stream)); t_lap[2] = now(); }
    std::unique_lock<std::mutex> lock(g_resident_launch, std::defer_lock);
    if (grid) {
        lock.lock();
        lcsgpu::NjLoopArgs p{};
        p.a = a;
        p.D2 = a.D + tri_floats;
        p.slots = (uint64_t*)(base + o_slots);
        p.u = (uint64_t*)(base + o_u);
        p.err = a.sel + 3;
        p.prof = profile ? (long long*)(base + o_slots + (size_t)grid * 32) : nullptr;
        p.cap = cap;
        p.compact_min = tune_int("nj_squeeze_min", 256);
        HIP_TRY(lcsgpu::launch_nj_loop(p, grid, L.stream));
        if (p.prof) {
            long long lap[6];
            HIP_TRY(hipMemcpyAsync(lap, p.prof, sizeof lap, hipMemcpyDeviceToHost, L.stream));
            HIP_TRY(hipStreamSynchronize(L.stream));
            t_lap[3] = now();
            fprintf(stderr, "lcsgpu_nj: LCS triangle ready -> buffers %.1f ms, float distances %.1f ms, initial sums + the launch %.1f ms\n",
                    (t_lap[1] - t_lap[0]) * 1e3, (t_lap[2] - t_lap[1]) * 1e3, (t_lap[3] - t_lap[2]) * 1e3);
            fprintf(stderr, "lcsgpu_nj: one resident launch of %d workgroups for %d merges; workgroup 0: squeezes %.1f ms, scan + chain %.1f, "
                            "(the chain alone %.1f), exchange %.1f, updates %.1f, polls %.1f\n", grid, n - 2, lap[0] * 1e-5, lap[1] * 1e-5,
                    lap[5] * 1e-5, lap[2] * 1e-5, lap[3] * 1e-5, lap[4] * 1e-5);
        }
    } else {
        HIP_TRY(lcsgpu::launch_nj(a, L.stream));
    }
    int32_t sel[4] = {0, 0, 0, 0};
    rc = fetch_merges(L, base + o_sel, 16, o_left - o_sel, o_right - o_sel, n, sel, out_left, out_right);
    if (rc) return rc;
    if (profile) fprintf(stderr, "lcsgpu_nj: results on the host %.1f ms after the launch was over (%.1f ms since the triangle was ready)\n", (now() - t_lap[3]) * 1e3, (now() - t_lap[0]) * 1e3);
    L.plan_in_flight = false;
    note_async_call(ctx);
    if (sel[3]) { // the workgroups of the resident launch did not meet (the CUs were not theirs alone): as launches, from the LCS values
        if (profile || getenv("LCSGPU_VERBOSE")) fprintf(stderr, "lcsgpu_nj: the resident launch gave up waiting; the merges run as launches\n");
        lock.unlock();
        return nj_reduce(ctx, L, elem, distance_kind, out_left, out_right, false);
    }
    if (sel[2])
        return fail(LCSGPU_E_INVALID, "NJ: no finite q (a pair with LCS 0?) -- the reference's result is degenerate "
                                      "for this input");
    return LCSGPU_OK;
}

} // namespace

extern "C" {

int lcsgpu_multi_upgma(lcsgpu_ctx* const* ctxs, int32_t n_ctx, int distance_kind, int modified, int32_t* out_left,
                       int32_t* out_right)
{
    int rc = check_multi(ctxs, n_ctx);
    if (rc) return rc;
    if (!valid_kind(distance_kind)) return fail(LCSGPU_E_INVALID, "unknown distance kind %d", distance_kind);
    const int32_t n = ctxs[0]->n;
    if (n < 2) return LCSGPU_OK;
    if (!out_left || !out_right) return fail(LCSGPU_E_INVALID, "NULL output");
    MultiGuard g(ctxs, n_ctx);
    const int elem = ctxs[0]->max_len > 65535 ? 4 : 2;
    if (n_ctx == 1) // one GPU: the triangle is computed block by block inside (2 B per pair never exist all at once)
        return upgma_reduce(ctxs[0], *g.lanes[0], elem, distance_kind, modified, out_left, out_right, false);
    rc = whole_triangle(ctxs, n_ctx, g, elem);
    if (rc) return rc;
    return upgma_reduce(ctxs[0], *g.lanes[0], elem, distance_kind, modified, out_left, out_right, true);
}

int lcsgpu_upgma(lcsgpu_ctx* ctx, int distance_kind, int modified, int32_t* out_left, int32_t* out_right)
{
    return lcsgpu_multi_upgma(&ctx, 1, distance_kind, modified, out_left, out_right);
}

int lcsgpu_multi_nj(lcsgpu_ctx* const* ctxs, int32_t n_ctx, int distance_kind, int32_t* out_left, int32_t* out_right)
{
    int rc = check_multi(ctxs, n_ctx);
    if (rc) return rc;
    if (!valid_kind(distance_kind)) return fail(LCSGPU_E_INVALID, "unknown distance kind %d", distance_kind);
    const int32_t n = ctxs[0]->n;
    if (n < 2) return LCSGPU_OK;
    if (!out_left || !out_right) return fail(LCSGPU_E_INVALID, "NULL output");
    const bool profile = getenv("LCSGPU_PROFILE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    MultiGuard g(ctxs, n_ctx);
    const auto t1 = std::chrono::steady_clock::now();
    const int elem = ctxs[0]->max_len > 65535 ? 4 : 2;
    rc = whole_triangle(ctxs, n_ctx, g, elem);
    if (rc) return rc;
    if (profile) {
        const auto t2 = std::chrono::steady_clock::now();
        (void)hipStreamSynchronize(g.lanes[0]->stream);
        const auto t3 = std::chrono::steady_clock::now();
        fprintf(stderr, "lcsgpu_nj: lanes %.1f ms, the LCS triangle: host side %.1f ms, then %.1f ms until the device is done\n",
                std::chrono::duration<double>(t1 - t0).count() * 1e3, std::chrono::duration<double>(t2 - t1).count() * 1e3,
                std::chrono::duration<double>(t3 - t2).count() * 1e3);
    }
    rc = nj_reduce(ctxs[0], *g.lanes[0], elem, distance_kind, out_left, out_right);
    if (profile)
        fprintf(stderr, "lcsgpu_nj: the call %.1f ms\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e3);
    return rc;
}

int lcsgpu_nj(lcsgpu_ctx* ctx, int distance_kind, int32_t* out_left, int32_t* out_right)
{
    return lcsgpu_multi_nj(&ctx, 1, distance_kind, out_left, out_right);
}

// Single linkage over several GPUs inside one process: every context computes its row block and the local half
// of each Boruvka round on its own GPU (all asynchronous, so the GPUs work at the same time).  The exchange stays
// in device memory: context k pushes its n keys (16 B each) into slot k of EVERY context's gathered buffer -- the
// all-gather of include/lcsgpu.h's protocol, written as N x (N-1) peer copies of n x 16 B on the producers' streams
// (xGMI; 1.6 MB each at n = 100 000) -- every context waits for the N pushes on its own stream (events, no host
// involvement) and runs the same global half (lcsgpu_mst_shard_merge's kernels), so the component labels stay
// replicated without being sent.  The host synchronises ONCE per round, for the edge count that ends the loop.
// The gathered buffers alternate between two halves by round parity: a producer's push of round r+2 is ordered
// after its own merge of round r+1, which waited for every context's push of round r+1, which in turn follows that
// context's merge of round r in stream order -- so no half is overwritten while a merge still reads it.
int lcsgpu_multi_mst_prim(lcsgpu_ctx* const* ctxs, int32_t n_ctx, int distance_kind, lcsgpu_mst_edge* out_edges)
{
    int rc = check_multi(ctxs, n_ctx);
    if (rc) return rc;
    const Exchange wanted = exchange_wanted(); // rccl asked for by name: also with one context, so that a 1-GPU box runs the calls
    if (n_ctx == 1 && wanted != Exchange::RCCL) return lcsgpu_mst_prim(ctxs[0], distance_kind, out_edges);
    const bool triangle_orientation = (distance_kind & LCSGPU_MST_TRIANGLE_ORIENTATION) != 0;
    const int kind = distance_kind & ~LCSGPU_MST_TRIANGLE_ORIENTATION;
    if (!valid_kind(kind)) return fail(LCSGPU_E_INVALID, "unknown distance kind %d", kind);
    const int32_t n = ctxs[0]->n;
    if (n < 2) return LCSGPU_OK;
    if (!out_edges) return fail(LCSGPU_E_INVALID, "NULL out_edges");
    if (!triangle_orientation)
        for (int32_t i = 0; i < n; ++i)
            if (ctxs[0]->quirk[i])
                return fail(LCSGPU_E_UNSUPPORTED, "sequence %d is orientation sensitive: MSTPrim's distances depend on which "
                                                  "endpoint is the ref (use lcsgpu_mst_prim on one context)", i);
    MultiGuard g(ctxs, n_ctx);
    // Whatever way this call ends, nothing of it may still run when its events are destroyed and the lanes are handed
    // back, and no context may be left pointing at a triangle block that the next call reuses.
    std::vector<hipEvent_t> pushed((size_t)2 * n_ctx, nullptr);
    struct Cleanup {
        lcsgpu_ctx* const* ctxs;
        int32_t n_ctx;
        MultiGuard& g;
        std::vector<hipEvent_t>& ev;
        ~Cleanup()
        {
            for (int k = 0; k < n_ctx; ++k) {
                (void)hipSetDevice(ctxs[k]->device);
                (void)hipStreamSynchronize(g.lanes[k]->stream);
                g.lanes[k]->plan_in_flight = false;
                ctxs[k]->mst.active = false;
            }
            for (hipEvent_t e : ev)
                if (e) (void)hipEventDestroy(e);
            (void)hipGetLastError();
        }
    } cleanup{ctxs, n_ctx, g, pushed};

    const int elem = ctxs[0]->max_len > 65535 ? 4 : 2;
    const std::vector<int32_t> cut = equal_pair_cuts(0, n, n_ctx);
    const size_t key_bytes = (size_t)n * sizeof(lcsgpu_mst_key);
    std::vector<int> devices(n_ctx);
    std::vector<hipStream_t> streams(n_ctx);
    bool distinct = true;
    for (int k = 0; k < n_ctx; ++k) {
        devices[k] = ctxs[k]->device;
        streams[k] = g.lanes[k]->stream;
        for (int j = 0; j < k; ++j) distinct = distinct && devices[j] != devices[k];
    }
    // the exchange of this call (see the comment above struct Rccl)
    RcclComms* rc_comms = nullptr;
    std::string note;
    if (wanted == Exchange::RCCL) {
        if ((rc = rccl_comms(devices, &rc_comms))) return rc;
        note = "rccl: one grouped ncclAllGather per round (LCSGPU_EXCHANGE=rccl)";
    } else if (wanted == Exchange::AUTO && n_ctx > 1 && distinct) {
        if (rccl_comms(devices, &rc_comms) == LCSGPU_OK) note = "rccl: one grouped ncclAllGather per round (automatic: every context on its own device)";
        else {
            rc_comms = nullptr;
            note = std::string("peer copies, because RCCL is not usable: ") + lcsgpu_last_error();
        }
    } else
        note = wanted == Exchange::PEER ? "peer copies (LCSGPU_EXCHANGE=peer)" : "peer copies (automatic: contexts share a device, RCCL wants one device per rank)";
    const std::vector<ncclComm_t>* comms = rc_comms ? &rc_comms->comm : nullptr;
    const char* chk = getenv("LCSGPU_EXCHANGE_CHECK");
    const bool cross_check = comms && !(chk && !strcmp(chk, "0")) && (!rc_comms->verified || (chk && !strcmp(chk, "always")));
    {
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        g_exchange_note = note;
    }
    if (getenv("LCSGPU_PROFILE")) fprintf(stderr, "lcsgpu_multi_mst_prim: %d contexts, key exchange by %s%s\n", n_ctx, note.c_str(),
                                          cross_check ? " -- first round cross-checked against peer copies" : "");
    for (int k = 0; k < n_ctx; ++k) {
        Lane& L = *g.lanes[k];
        const int32_t r0 = cut[k], r1 = cut[k + 1];
        const int64_t off = tri_offset(r0);
        // the block's triangle stays in this GPU's HBM when it fits; else nothing is kept and every round recomputes the
        // block with the fold fused into the launch (lcsgpu_mst_shard_begin, LCSGPU_MST_COMPUTE); LCSGPU_MST_MODE as above
        const char* mode = getenv("LCSGPU_MST_MODE");
        const bool want_fused = mode && !strcmp(mode, "fused") && elem == 2;
        bool resident = !(mode && !strcmp(mode, "recompute") && elem == 2);
        if (resident) {
            rc = reserve_big(ctxs[k], L.d_out, (size_t)std::max<int64_t>(tri_offset(r1) - off, 1) * elem, "a row block of the LCS triangle");
            if (rc == LCSGPU_E_NOMEM && elem == 2) resident = false;
            else if (rc) return rc;
        }
        HIP_TRY(hipSetDevice(ctxs[k]->device));
        if (!resident || want_fused)
            rc = shard_begin(ctxs[k], L, resident ? L.d_out.p : nullptr, elem, r0, r1, kind, true);
        else {
            if (r1 > r0) {
                rc = run_rows(ctxs[k], L, lcsgpu::MODE_TRIANGLE, nullptr, r0, r1 - r0, nullptr, 0, std::max(0, r1 - 1), L.d_out.p, 0, off, elem, r0);
                if (rc) return rc;
            }
            rc = shard_begin(ctxs[k], L, L.d_out.p, elem, r0, r1, kind);
        }
        if (rc) return rc;
        HIP_TRY(ctxs[k]->d_gather.reserve(2 * (size_t)n_ctx * key_bytes));
        for (int h = 0; h < 2; ++h) HIP_TRY(hipEventCreateWithFlags(&pushed[(size_t)h * n_ctx + k], hipEventDisableTiming));
    }
    int32_t found = 0;
    for (int round = 0; found < n - 1; ++round) {
        if (round > 64) return fail(LCSGPU_E_STATE, "MST: the Boruvka rounds do not converge");
        const int half = round & 1;
        const bool check_now = cross_check && round == 0;
        for (int k = 0; k < n_ctx; ++k) { // local halves everywhere, each followed by its pushes
            HIP_TRY(hipSetDevice(ctxs[k]->device));
            char* own = (char*)ctxs[k]->d_gather.p + ((size_t)half * n_ctx + k) * key_bytes;
            rc = shard_best(ctxs[k], *g.lanes[k], own, nullptr); // straight into its own slot
            if (rc) return rc;
            if (comms && !check_now) continue; // the exchange is one grouped all-gather below
            // peer copies: the exchange itself -- or, in the cross-checked round, the second opinion, into the idle half
            const int to_half = comms ? half ^ 1 : half;
            for (int j = 0; j < n_ctx; ++j) {
                if (j == k && !comms) continue;
                char* slot = (char*)ctxs[j]->d_gather.p + ((size_t)to_half * n_ctx + k) * key_bytes;
                rc = copy_between(ctxs[j], slot, ctxs[k], own, key_bytes, g.lanes[k]->stream);
                if (rc) return rc;
            }
            HIP_TRY(hipSetDevice(ctxs[k]->device));
            HIP_TRY(hipEventRecord(pushed[(size_t)half * n_ctx + k], g.lanes[k]->stream));
        }
        if (comms) {
            std::vector<char*> base(n_ctx);
            for (int k = 0; k < n_ctx; ++k) base[k] = (char*)ctxs[k]->d_gather.p + (size_t)half * n_ctx * key_bytes;
            rc = rccl_all_gather_keys(*comms, base, key_bytes, streams, devices);
            if (rc) return rc;
        }
        if (check_now) {
            // both forms have been queued: wait for every producer, then compare the two halves of every context
            for (int k = 0; k < n_ctx; ++k) {
                HIP_TRY(hipSetDevice(ctxs[k]->device));
                HIP_TRY(hipStreamSynchronize(g.lanes[k]->stream));
            }
            std::vector<char> by_rccl((size_t)n_ctx * key_bytes), by_peer((size_t)n_ctx * key_bytes);
            for (int j = 0; j < n_ctx; ++j) {
                HIP_TRY(hipSetDevice(ctxs[j]->device));
                HIP_TRY(hipMemcpy(by_rccl.data(), (char*)ctxs[j]->d_gather.p + (size_t)half * n_ctx * key_bytes, by_rccl.size(), hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(by_peer.data(), (char*)ctxs[j]->d_gather.p + (size_t)(half ^ 1) * n_ctx * key_bytes, by_peer.size(), hipMemcpyDeviceToHost));
                for (int k = 0; k < n_ctx; ++k)
                    if (memcmp(by_rccl.data() + (size_t)k * key_bytes, by_peer.data() + (size_t)k * key_bytes, key_bytes))
                        return fail(LCSGPU_E_HIP, "the RCCL all-gather and the peer copies disagree: context %d (device %d) holds different keys "
                                                  "of context %d (device %d) in the two forms (%s) -- LCSGPU_EXCHANGE=peer selects the peer copies",
                                    j, devices[j], k, devices[k], transport_between(devices[k], devices[j]));
            }
            std::lock_guard<std::mutex> lk(g_rccl_mu);
            rc_comms->verified = true;
        }
        for (int j = 0; j < n_ctx; ++j) { // global halves: each context over all N slots, once they have landed
            HIP_TRY(hipSetDevice(ctxs[j]->device));
            for (int k = 0; k < n_ctx && !comms; ++k) // (the all-gather is ordered on the lane's own stream)
                if (k != j) HIP_TRY(hipStreamWaitEvent(g.lanes[j]->stream, pushed[(size_t)half * n_ctx + k], 0));
            rc = shard_merge_async(ctxs[j], *g.lanes[j], (char*)ctxs[j]->d_gather.p + (size_t)half * n_ctx * key_bytes, n_ctx);
            if (rc) return rc;
        }
        HIP_TRY(hipSetDevice(ctxs[0]->device));
        rc = shard_count(ctxs[0], *g.lanes[0], &found); // the round's one host synchronisation
        if (rc) return rc;
        for (int k = 1; k < n_ctx; ++k) ctxs[k]->mst.found = found; // same keys, same kernels: same count everywhere
    }
    // The loop watched context 0 only.  Before its edge list is taken for the answer: every other context must have
    // run to the end without a HIP error and without the merge kernels' inconsistent-keys flag, and must have counted
    // the same edges -- a failed push, all-gather or merge on another GPU would otherwise pass unseen.
    for (int k = 1; k < n_ctx; ++k) {
        HIP_TRY(hipSetDevice(ctxs[k]->device));
        int32_t c[2] = {0, 0};
        hipError_t e = hipMemcpyAsync(c, ctxs[k]->mst.b.counters, 8, hipMemcpyDeviceToHost, g.lanes[k]->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(g.lanes[k]->stream);
        if (e != hipSuccess) return fail(LCSGPU_E_HIP, "context %d (device %d) failed during the Boruvka rounds: %s", k, devices[k], hipGetErrorString(e));
        g.lanes[k]->plan_in_flight = false;
        if (c[1]) return fail(LCSGPU_E_STATE, "MST: context %d (device %d) saw inconsistent keys (a hooking cycle)", k, devices[k]);
        if (c[0] != found)
            return fail(LCSGPU_E_STATE, "MST: context %d (device %d) recorded %d edges, context 0 recorded %d -- the contexts did not "
                                        "see the same keys (exchange: %s)", k, devices[k], c[0], found, note.c_str());
    }
    HIP_TRY(hipSetDevice(ctxs[0]->device));
    rc = shard_finish(ctxs[0], *g.lanes[0], out_edges); // context 0's copy of the edge list, in Prim's order
    if (rc) return rc;
    note_async_call(ctxs[0]);
    for (int k = 1; k < n_ctx; ++k) note_async_call(ctxs[k], true); // lcsgpu_last_kernel_ms answers for every context of the call
    return LCSGPU_OK;
}

// How the contexts of a list reach each other, as text (one line per fact, '\n' separated, NUL terminated, truncated to
// cap): the transport copy_between would pick for every ordered pair of contexts -- counted per kind, with the pairs that
// do NOT get a peer copy named -- and the key exchange the last lcsgpu_multi_mst_prim of this process used and why.
// Asking switches peer access on for the pairs (as the first copy would).
int lcsgpu_multi_transport(lcsgpu_ctx* const* ctxs, int32_t n_ctx, char* buf, size_t cap)
{
    if (!ctxs || n_ctx < 1 || !buf || cap < 2) return fail(LCSGPU_E_INVALID, "bad argument");
    for (int k = 0; k < n_ctx; ++k)
        if (!ctxs[k]) return fail(LCSGPU_E_INVALID, "NULL ctx");
    std::map<std::string, int> kinds;
    std::string odd;
    for (int a = 0; a < n_ctx; ++a)
        for (int b = 0; b < n_ctx; ++b) {
            if (a == b) continue;
            const char* t = transport_between(ctxs[a]->device, ctxs[b]->device);
            ++kinds[t];
            if (strcmp(t, "peer-copy") && strcmp(t, "same-device") && odd.size() < 400) {
                char one[96];
                snprintf(one, sizeof one, " %d->%d(dev %d->%d)", a, b, ctxs[a]->device, ctxs[b]->device);
                odd += one;
            }
        }
    std::string text = "contexts: " + std::to_string(n_ctx) + " on devices";
    for (int k = 0; k < n_ctx; ++k) text += " " + std::to_string(ctxs[k]->device);
    text += "\ncopies between contexts:";
    if (kinds.empty()) text += " none (one context)";
    for (const auto& kv : kinds) text += " " + kv.first + " x" + std::to_string(kv.second);
    if (!odd.empty()) text += "\nnot by peer copy:" + odd;
    {
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        text += "\nkey exchange of the last single-linkage call: " + g_exchange_note;
    }
    snprintf(buf, cap, "%s", text.c_str());
    return LCSGPU_OK;
}

// Rows [row_begin, row_end) of the lower triangle into HOST memory, the rows split into blocks of equal pair
// counts, one per context.
int lcsgpu_multi_lcs_triangle(lcsgpu_ctx* const* ctxs, int32_t n_ctx, int32_t row_begin, int32_t row_end, void* out, int elem_size)
{
    int rc = check_multi(ctxs, n_ctx);
    if (rc) return rc;
    if (row_begin < 0 || row_end < row_begin || row_end > ctxs[0]->n) return fail(LCSGPU_E_INVALID, "bad row range");
    if (elem_size != 2 && elem_size != 4) return fail(LCSGPU_E_INVALID, "elem_size must be 2 or 4");
    const int64_t base = tri_offset(row_begin), count = tri_offset(row_end) - base;
    if (count <= 0) return LCSGPU_OK;
    if (!out) return fail(LCSGPU_E_INVALID, "NULL out");
    if (n_ctx == 1 || count < (1 << 20)) return lcsgpu_lcs_triangle(ctxs[0], row_begin, row_end, out, elem_size);
    const std::vector<int32_t> cut = equal_pair_cuts(row_begin, row_end, n_ctx);
    MultiGuard g(ctxs, n_ctx);
    for (int k = 0; k < n_ctx; ++k) {
        const int32_t r0 = cut[k], r1 = cut[k + 1];
        if (r1 <= r0) continue;
        Lane& L = *g.lanes[k];
        const int64_t off = tri_offset(r0);
        const size_t bytes = (size_t)(tri_offset(r1) - off) * elem_size;
        rc = reserve_big(ctxs[k], L.d_out, bytes, "a row block of the LCS triangle");
        if (rc) return rc;
        rc = run_rows(ctxs[k], L, lcsgpu::MODE_TRIANGLE, nullptr, r0, r1 - r0, nullptr, 0, std::max(0, r1 - 1), L.d_out.p, 0, off, elem_size, r0);
        if (rc) return rc;
    }
    // Every GPU is computing by now.  Each block leaves over its own GPU's PCIe link, all links at the same time: one
    // host thread per context (a copy into pageable host memory is staged by the runtime on the calling thread, so
    // queueing the copies from ONE thread would still drain the GPUs one after the other -- at 100 000 sequences 10 GB
    // over one link, ~0.4 s, instead of 1.25 GB over each of eight).
    std::vector<int> status(n_ctx, LCSGPU_OK);
    std::vector<std::string> what(n_ctx);
    auto drain = [&](int k) {
        const int32_t r0 = cut[k], r1 = cut[k + 1];
        if (r1 <= r0) return;
        Lane& L = *g.lanes[k];
        const int64_t off = tri_offset(r0);
        hipError_t e = hipSetDevice(ctxs[k]->device);
        if (e == hipSuccess)
            e = hipMemcpyAsync((char*)out + (size_t)(off - base) * elem_size, L.d_out.p, (size_t)(tri_offset(r1) - off) * elem_size,
                               hipMemcpyDeviceToHost, L.stream);
        if (e == hipSuccess) e = hipStreamSynchronize(L.stream);
        if (e != hipSuccess) {
            status[k] = LCSGPU_E_HIP;
            what[k] = hipGetErrorString(e);
        }
    };
    {
        std::vector<std::thread> workers;
        for (int k = 1; k < n_ctx; ++k) workers.emplace_back(drain, k);
        drain(0);
        for (auto& t : workers) t.join();
    }
    for (int k = 0; k < n_ctx; ++k) {
        if (status[k]) return fail(status[k], "row block %d (device %d) did not arrive: %s", k, ctxs[k]->device, what[k].c_str());
        if (cut[k + 1] > cut[k]) finish_host_call(ctxs[k], *g.lanes[k]);
    }
    return LCSGPU_OK;
}

} // extern "C"
