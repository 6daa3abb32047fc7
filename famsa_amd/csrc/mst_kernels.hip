// mst_kernels.hip -- the minimum spanning tree of the uploaded set by Boruvka rounds over ROW BLOCKS of
// the LCS triangle in HBM: one GPU holding the whole triangle, or the GPUs of a node holding one row
// block each (the N x N pair space tiled by row block).
//
// MSTPrim (reference tree/MSTPrim.cpp:356-533) orders edges by the strict total order
// (d, ~pack(min id, max id)) -- smaller distance first, then the larger packed id (cpp:493-509) -- so the
// MST is unique and any exact MST algorithm yields the reference's edge set; Prim's insertion order from
// vertex 0 (cpp:372-391) is then a walk over those n-1 edges, done by the caller on the host, and the
// dendrogram (cpp:784-833) follows from that order.  Prim itself needs n-1 dependent steps; Boruvka needs
// <= log2(n) rounds of streaming passes (2 B per pair).  A round has a LOCAL half and a GLOBAL half:
//   local  (per row block [r0, r1), no communication): every vertex's best edge to another component
//          among the pairs the block holds -- the row part (u < v, v in the block: v's own contiguous row)
//          and the column part (u > v, u in the block: lanes = consecutive columns, walking down the
//          rows, coalesced) -> best[v], n x 16 B.  Round 0 of this is the "per-row minima" of the
//          north star, completed by the column part.
//   global (replicated on every GPU after the exchange of the best[] arrays -- n x 16 B per GPU, an
//          all-gather): fold the blocks' keys, every component's best edge (two 64-bit atomic-min phases:
//          distance bits, then id), hook each component to the other end of its edge (mutual choices:
//          the smaller root stays a root and the edge is recorded once), relabel the vertices.
// The component labels are a pure function of the exchanged keys, so every GPU derives the same ones.
//
// Distances: Transform<double, kind> (reference tree/AbstractTreeGenerator.hpp:28-82) exactly -- host-built
// pow table + IEEE f64 division -- but only for candidates that can win: a multiplication-only float test
// (certainly_worse) against a per-lane threshold kept just above the lane's current exact best proves most
// candidates larger than the best, and the table look-up, the f64 division and the 128-bit compare are
// skipped for them.  History of the two passes at n = 100 000 (10 GB of u16 each) is in DESIGN.md.
//
// Valid when d(u, v) does not depend on which endpoint is the ref: always for the triangle's own
// orientation (what SLINK sees), and for MSTPrim's orientation when no uploaded sequence is orientation
// sensitive (SURVEY note Q); otherwise the caller keeps the Prim kernel.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "lcs_kernels.h"

namespace lcsgpu {

namespace {

constexpr unsigned long long NO_D = 0x7fefffffffffffffull; // bits of DBL_MAX: "no candidate"
constexpr unsigned long long NO_ID = ~0ull;

__device__ __forceinline__ bool key_less(unsigned long long d1, unsigned long long i1, unsigned long long d2,
                                         unsigned long long i2)
{
    return d1 < d2 || (d1 == d2 && i1 < i2); // distances are >= 0: their bit patterns order like the values
}

__device__ __forceinline__ unsigned long long pack_ids(uint32_t a, uint32_t b) // ids_to_uint64, tree/MSTPrim.h:432-439
{
    return a < b ? ((unsigned long long)a << 32) + b : ((unsigned long long)b << 32) + a;
}

// a lane's running best: exact key, the (l, indel) it came from, and the float threshold that admits every
// candidate able to beat it (KIND 1: the threshold lives in the d^4 domain, see certainly_worse)
struct Best {
    unsigned long long d = NO_D, id = NO_ID;
    uint32_t l = ~0u, indel = ~0u;
    float thr = __builtin_inff();
};

// pow(i, 0.75) for the exact path: from LDS when the table fits -- the exact path then touches no global
// memory, so it never waits for the triangle loads in flight (and the compiler's s_waitcnt vmcnt bookkeeping
// of those loads stays exact) -- else from HBM
template <bool IN_LDS>
struct PowTable {
    const double* p;
    __device__ __forceinline__ double operator()(uint32_t i) const { return p[i]; }
};
template <bool IN_LDS>
__device__ __forceinline__ PowTable<IN_LDS> stage_pow_table(const BoruvkaArgs& a, double* smem)
{
    if (!IN_LDS) return PowTable<IN_LDS>{a.pow_table};
    for (int i = threadIdx.x; i < a.pow_n; i += 256) smem[i] = a.pow_table[i];
    __syncthreads();
    return PowTable<IN_LDS>{smem};
}

// Pre-filter, multiplications only.  KIND 1: d = indel^0.75 / l, and d > t  <=>  indel^3 > t^4 l^4; KIND 0:
// d = indel / l > t  <=>  indel > t l.  `thr` holds (best d x (1 + 2^-14))^4 resp. best d x (1 + 2^-14), the
// five float roundings stay below 2^-21, so "greater" is certain.  l == 0 or an infinite threshold give
// NaN / inf on the right-hand side: not "greater", the exact path decides.  `excluded` (same component)
// counts as worse than anything.
template <int KIND>
__device__ __forceinline__ bool may_win(float thr, uint32_t l, uint32_t indel, bool excluded)
{
    const float x = (float)indel, lf = (float)l;
    float lhs, rhs;
    if (KIND == 1) {
        const float l2 = lf * lf;
        lhs = (x * x) * x;
        rhs = thr * (l2 * l2);
    } else {
        lhs = x;
        rhs = thr * lf;
    }
    lhs = excluded ? __builtin_inff() : lhs;
    return !(lhs > rhs);
}

// The exact comparison: Transform<double, KIND> (hpp:28-82) and MSTPrim's key order.  A candidate with the
// best's own (l, indel) has the best's distance bit for bit: only the ids decide, no division.  Returns true
// if the threshold changed.
template <int KIND, typename PW>
__device__ __forceinline__ bool exact_update(const PW& pw, Best& b, uint32_t l, uint32_t indel, uint32_t lo, uint32_t hi)
{
    const unsigned long long id = ~(((unsigned long long)lo << 32) + hi);
    if (l == b.l && indel == b.indel) {
        if (id < b.id) b.id = id;
        return false;
    }
    double d;
    if (l == 0) d = 1.7976931348623155e308; // nextafter(DBL_MAX, 0), hpp:61,73
    else if (KIND == 1) d = pw(indel) / (double)l;
    else d = (double)indel / (double)l;
    const unsigned long long db = (unsigned long long)__double_as_longlong(d);
    if (!key_less(db, id, b.d, b.id)) return false;
    b.d = db;
    b.id = id;
    b.l = l;
    b.indel = indel;
    const float t = __double2float_ru(d) * 1.00006104f; // x (1 + 2^-14), stays >= d (1 + 2^-15)
    if (KIND == 1) {
        const float t2 = t * t;
        float t4 = t2 * t2;
        if (t4 < 1e-30f && t != 0.0f) t4 = __builtin_inff(); // no float headroom left: everything goes the exact way
        b.thr = t4;
    } else {
        b.thr = t;
    }
    return true;
}

// smallest value over the wave, for non-negative floats (their bit patterns order like the values)
__device__ __forceinline__ float wave_min_nonneg(float f)
{
    uint32_t x = __float_as_uint(f);
    uint32_t y;
    y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0xB1, 0xF, 0xF, false);  x = y < x ? y : x; // quad_perm [1,0,3,2]
    y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x4E, 0xF, 0xF, false);  x = y < x ? y : x; // quad_perm [2,3,0,1]
    y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x141, 0xF, 0xF, false); x = y < x ? y : x; // row_half_mirror
    y = (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, 0x140, 0xF, 0xF, false); x = y < x ? y : x; // row_mirror
    uint32_t r = (uint32_t)__builtin_amdgcn_readlane((int)x, 0);
    y = (uint32_t)__builtin_amdgcn_readlane((int)x, 16); r = y < r ? y : r;
    y = (uint32_t)__builtin_amdgcn_readlane((int)x, 32); r = y < r ? y : r;
    y = (uint32_t)__builtin_amdgcn_readlane((int)x, 48); r = y < r ? y : r;
    return __uint_as_float(r);
}

} // namespace

__global__ __launch_bounds__(256) void boruvka_init_kernel(BoruvkaArgs a)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v < a.n) a.comp[v] = v;
    if (v == 0) a.counters[0] = 0; // edges recorded so far
}

// What bounds the two passes (PMC, n = 50 000): not HBM -- the CU's single SCALAR unit (80% busy: per-element
// index arithmetic, mask combining, branches and scalar-load addresses of 4 SIMDs' waves) and the exact path
// (a wave takes it when ANY lane has a candidate that may win: 30-70% of the elements while every lane keeps
// its own young threshold).  Hence: (1) the bulk of each pass runs without range tests, on batches whose
// per-row data come in wide scalar loads, with ONE branch per group of four elements; (2) ties with the
// current best are settled on the ids alone; (3) the row pass, where all lanes of a wave work for the same
// vertex, shares the threshold across the wave after every update; the column pass gives each lane a long
// stream (few row chunks) so thresholds mature early; (4) two batches of loads are in flight per lane, all of
// them unconditional (clamped indices), so the loop bodies are straight-line code with exact s_waitcnt counts.
constexpr int COLS_PER_WG = 256; // columns of one workgroup of the column pass
constexpr int ROWS_PER_WG = 4;   // rows of one workgroup of the row pass: they share the per-column loads
typedef int int4_a4 __attribute__((ext_vector_type(4), aligned(4)));

// the smallest key of the workgroup -> returned in thread 0 (all 256 threads call it)
__device__ __forceinline__ MstKey block_min_key(unsigned long long d, unsigned long long id, MstKey* s_wave /* [4] */)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long d2 = __shfl_xor(d, o, 64), i2 = __shfl_xor(id, o, 64);
        if (key_less(d2, i2, d, id)) { d = d2; id = i2; }
    }
    __syncthreads(); // s_wave may still be read by the previous call
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = MstKey{d, id};
    __syncthreads();
    MstKey k = s_wave[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (key_less(s_wave[w].d, s_wave[w].id, k.d, k.id)) k = s_wave[w];
    return k;
}

// best edge of the vertices v = r0 + ROWS_PER_WG * blockIdx.x + r among u < v (row v of the triangle), to
// another component; lanes stride over the columns.
template <typename T, int KIND, bool POW_LDS>
__global__ __launch_bounds__(256) void boruvka_row_kernel(BoruvkaArgs a)
{
    __shared__ MstKey s_wave[4];
    extern __shared__ double s_pow[];
    constexpr int R = ROWS_PER_WG, UNR = 4;
    const int tid = threadIdx.x;
    const PowTable<POW_LDS> pw = stage_pow_table<POW_LDS>(a, s_pow);
    const int vb = a.r0 + blockIdx.x * R;
    const int nr = min(R, a.r1 - vb);
    const int vmax = vb + nr - 1; // the longest row of the tile
    if (vmax < 1) { // row 0 alone: nothing below it
        if (tid == 0) a.row_best[vb] = MstKey{NO_D, NO_ID};
        return;
    }
    int v[R], cv[R], last[R];
    uint32_t len_v[R];
    const T* row[R];
    Best b[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int vr = r < nr ? vb + r : vmax;
        cv[r] = a.comp[vr];
        len_v[r] = a.lens[vr];
        const int vp = vr >= 1 ? vr : vmax;        // the row the loads go to (row 0 has no elements)
        row[r] = (const T*)a.tri + ((int64_t)vp * (vp - 1) / 2 - a.off);
        last[r] = vp - 1;                            // its last valid column
        v[r] = r < nr ? vr : 0;                      // columns u < v[r] count (0: none)
        if (r >= nr) b[r].thr = -1.0f;               // a missing row: every candidate is "certainly worse"
    }
    T l[2][R][UNR];
    int cu[2][UNR];
    uint32_t lu[2][UNR];
    auto request = [&](int ub, int s) {
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const int u = ub + 256 * k;
            const int uc = min(u, vmax);
            cu[s][k] = a.comp[uc];
            lu[s][k] = a.lens[uc];
#pragma unroll
            for (int r = 0; r < R; ++r) l[s][r][k] = row[r][min(u, last[r])];
        }
    };
    // BULK = true: every column of the batch lies below every row of the tile (u < vb): no range tests
    auto evaluate = [&](int ub, int s, auto bulk) {
        constexpr bool BULK = decltype(bulk)::value;
        bool changed = false;
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const int u = ub + 256 * k;
            bool pass[R];
            uint32_t indel[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const uint32_t lv = l[s][r][k];
                indel[r] = len_v[r] + lu[s][k] - 2u * lv;
                pass[r] = may_win<KIND>(b[r].thr, lv, indel[r], cu[s][k] == cv[r] || (!BULK && u >= v[r]));
            }
            bool any = false;
#pragma unroll
            for (int r = 0; r < R; ++r) any |= pass[r];
            if (any) {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (pass[r] && r < nr && u < v[r] && cu[s][k] != cv[r])
                        changed |= exact_update<KIND>(pw, b[r], l[s][r][k], indel[r], (uint32_t)u, (uint32_t)v[r]);
            }
        }
        // all lanes of the wave work for the same vertices: a threshold one lane has reached holds for all
        if (__builtin_amdgcn_ballot_w64(changed) != 0) {
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (r < nr) b[r].thr = wave_min_nonneg(b[r].thr);
        }
    };
    constexpr int STEP = 256 * UNR;
    // batches [ub, ub + STEP) with ub + STEP <= vb lie entirely below the tile's rows; processed in pairs
    int ub = tid;
    const int n_pairs = vb / (2 * STEP);
    request(ub, 0);
    for (int it = 0; it < n_pairs; ++it, ub += 2 * STEP) {
        request(ub + STEP, 1);
        evaluate(ub, 0, std::true_type{});
        request(ub + 2 * STEP, 0);
        evaluate(ub + STEP, 1, std::true_type{});
    }
    for (; ub < vmax; ub += 2 * STEP) { // the rest, with range tests (l[0] holds the batch at ub)
        request(ub + STEP, 1);
        evaluate(ub, 0, std::false_type{});
        request(ub + 2 * STEP, 0);
        evaluate(ub + STEP, 1, std::false_type{});
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const MstKey k = block_min_key(b[r].d, b[r].id, s_wave);
        if (tid == 0 && r < nr) a.row_best[vb + r] = k;
    }
}

// best edge of vertex v (lane = column) among the rows u > v of one row chunk of the block -> part[chunk][v].
// The row index is uniform across the workgroup: comp[u] / lens[u] are scalar loads, 16 rows per load.
template <typename T, int KIND, bool POW_LDS>
__global__ __launch_bounds__(256) void boruvka_col_kernel(BoruvkaArgs a)
{
    constexpr int UNR = 16, GRP = 4; // rows per batch (two batches in flight); elements per branch
    const int c0 = blockIdx.x * COLS_PER_WG;
    const int v = c0 + threadIdx.x;
    const int chunk = blockIdx.y;
    const int u0 = a.r0 + chunk * a.rows_per_chunk, u1 = min(a.r1, u0 + a.rows_per_chunk);
    if (u0 >= u1 || c0 + 1 >= u1) return; // empty chunk, or every column of this workgroup lies at or above its last row: fold skips it
    extern __shared__ double s_pow[];
    const PowTable<POW_LDS> pw = stage_pow_table<POW_LDS>(a, s_pow);
    if (v >= a.n) return;
    Best b;
    const int cv = a.comp[v];
    const uint32_t len_v = a.lens[v];
    const T* tri = (const T*)a.tri - a.off;

    // rows [ua, ub): one by one, with the range test (the diagonal block and the chunk's last rows)
    auto plain_rows = [&](int ua, int ub) {
        for (int u = ua; u < ub; ++u) {
            const int cu = a.comp[u];
            const uint32_t len_u = a.lens[u];
            if (u > v && cu != cv) {
                const uint32_t lv = tri[(int64_t)u * (u - 1) / 2 + v], indel = len_u + len_v - 2u * lv;
                if (may_win<KIND>(b.thr, lv, indel, false)) exact_update<KIND>(pw, b, lv, indel, (uint32_t)v, (uint32_t)u);
            }
        }
    };
    const int first = max(u0, c0 + 1);
    const int bulk0 = min(u1, max(first, c0 + COLS_PER_WG)); // from here on every lane's column lies below the row
    plain_rows(first, bulk0);

    T l[2][UNR];
    auto request = [&](int ub, int s) {
#pragma unroll
        for (int k = 0; k < UNR; ++k) {
            const int64_t us = min(ub + k, u1 - 1); // > v here
            l[s][k] = tri[us * (us - 1) / 2 + v];
        }
    };
    auto evaluate = [&](int ub, int s) { // a full batch of the bulk: rows ub .. ub + UNR - 1 < u1, all below... above every lane's column
        int cu[UNR];
        uint32_t len_u[UNR];
#pragma unroll
        for (int q = 0; q < UNR / 4; ++q) {
            const int4_a4 c = *(const int4_a4*)(a.comp + ub + 4 * q);
            const int4_a4 n = *(const int4_a4*)(a.lens + ub + 4 * q);
            cu[4 * q] = c.x; cu[4 * q + 1] = c.y; cu[4 * q + 2] = c.z; cu[4 * q + 3] = c.w;
            len_u[4 * q] = (uint32_t)n.x; len_u[4 * q + 1] = (uint32_t)n.y; len_u[4 * q + 2] = (uint32_t)n.z; len_u[4 * q + 3] = (uint32_t)n.w;
        }
#pragma unroll
        for (int g = 0; g < UNR; g += GRP) {
            bool pass[GRP];
            uint32_t indel[GRP];
#pragma unroll
            for (int j = 0; j < GRP; ++j) {
                const uint32_t lv = l[s][g + j];
                indel[j] = len_u[g + j] + len_v - 2u * lv;
                pass[j] = may_win<KIND>(b.thr, lv, indel[j], cu[g + j] == cv);
            }
            bool any = false;
#pragma unroll
            for (int j = 0; j < GRP; ++j) any |= pass[j];
            if (any) {
#pragma unroll
                for (int j = 0; j < GRP; ++j)
                    if (pass[j] && cu[g + j] != cv)
                        exact_update<KIND>(pw, b, l[s][g + j], indel[j], (uint32_t)v, (uint32_t)(ub + g + j));
            }
        }
    };
    int ub = bulk0;
    const int n_pairs = (u1 - bulk0) / (2 * UNR);
    if (n_pairs > 0) {
        request(ub, 0);
        for (int it = 0; it < n_pairs; ++it, ub += 2 * UNR) {
            request(ub + UNR, 1);
            evaluate(ub, 0);
            request(ub + 2 * UNR, 0); // past the last pair: clamped rows, never evaluated
            evaluate(ub + UNR, 1);
        }
    }
    plain_rows(ub, u1);
    a.part[(size_t)chunk * a.n + v] = MstKey{b.d, b.id};
}

// this block's best edge per vertex: its row part (rows of the block) and the column partials that exist
__global__ __launch_bounds__(256) void boruvka_fold_kernel(BoruvkaArgs a)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= a.n) return;
    unsigned long long bd = NO_D, bi = NO_ID;
    if (v >= a.r0 && v < a.r1) {
        const MstKey k = a.row_best[v];
        bd = k.d;
        bi = k.id;
    }
    const int c0 = v / COLS_PER_WG * COLS_PER_WG;
    for (int c = 0; c < a.n_chunks; ++c) {
        const int u0 = a.r0 + c * a.rows_per_chunk, u1 = min(a.r1, u0 + a.rows_per_chunk);
        if (u0 >= u1 || c0 + 1 >= u1) continue; // not written (see boruvka_col_kernel)
        const MstKey k = a.part[(size_t)c * a.n + v];
        if (key_less(k.d, k.id, bd, bi)) { bd = k.d; bi = k.id; }
    }
    a.best[v] = MstKey{bd, bi};
}

__global__ __launch_bounds__(256) void boruvka_reset_kernel(BoruvkaArgs a)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= a.n) return;
    a.cb_d[v] = NO_D;
    a.cb_id[v] = NO_ID;
    a.parent[v] = v;
}

// fold the blocks' keys into vbest[v]; first atomic phase of the per-component minimum
__global__ __launch_bounds__(256) void boruvka_gather_kernel(BoruvkaArgs a, const MstKey* gathered, int n_parts)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= a.n) return;
    unsigned long long bd = NO_D, bi = NO_ID;
    for (int p = 0; p < n_parts; ++p) {
        const MstKey k = gathered[(size_t)p * a.n + v];
        if (key_less(k.d, k.id, bd, bi)) { bd = k.d; bi = k.id; }
    }
    a.vbest[v] = MstKey{bd, bi};
    // late rounds: thousands of vertices per component -- look first, most of them cannot lower the minimum
    // (a stale look only costs a redundant atomic)
    if (bi != NO_ID) {
        unsigned long long* slot = &a.cb_d[a.comp[v]];
        if (bd < __atomic_load_n(slot, __ATOMIC_RELAXED)) atomicMin(slot, bd);
    }
}

// second atomic phase: among the vertices that reach their component's smallest distance, the smallest id
__global__ __launch_bounds__(256) void boruvka_pick_kernel(BoruvkaArgs a)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= a.n) return;
    const MstKey k = a.vbest[v];
    if (k.id == NO_ID) return;
    const int c = a.comp[v];
    if (k.d == a.cb_d[c] && k.id < __atomic_load_n(&a.cb_id[c], __ATOMIC_RELAXED)) atomicMin(&a.cb_id[c], k.id);
}

// every component root hooks itself to the component at the other end of its edge
__global__ __launch_bounds__(256) void boruvka_hook_kernel(BoruvkaArgs a)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= a.n || a.comp[c] != c) return;
    const unsigned long long id = a.cb_id[c];
    if (id == NO_ID) return; // the last component
    const unsigned long long packed = ~id;
    const int x = (int)(packed >> 32), y = (int)(packed & 0xffffffffull);
    const int cx = a.comp[x], cy = a.comp[y];
    const int other = cx == c ? cy : cx;
    a.parent[c] = other;
    // the edge is recorded once: by its only chooser, or by the smaller of two components that chose each other
    const bool mutual = a.cb_id[other] == id;
    if (!mutual || c < other) {
        const int at = atomicAdd(&a.counters[0], 1);
        if (at >= a.n - 1) return; // cannot happen with consistent keys; the host reports the count
        a.edges[at].from = x;
        a.edges[at].to = y;
        a.edges[at].dist = __longlong_as_double((long long)a.cb_d[c]);
    }
}

// two components that chose each other form a 2-cycle: the smaller one becomes the root
__global__ __launch_bounds__(256) void boruvka_uncycle_kernel(BoruvkaArgs a)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= a.n || a.comp[c] != c) return;
    const int p = a.parent[c];
    if (p != c && a.parent[p] == c && c < p) a.parent[c] = c;
}

__global__ __launch_bounds__(256) void boruvka_relabel_kernel(BoruvkaArgs a)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= a.n) return;
    int r = a.comp[v];
    for (int p = a.parent[r]; p != r; p = a.parent[r]) r = p;
    a.comp_next[v] = r;
}

hipError_t launch_boruvka_init(const BoruvkaArgs& a, hipStream_t stream)
{
    hipLaunchKernelGGL(boruvka_init_kernel, dim3((a.n + 255) / 256), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_boruvka_best(const BoruvkaArgs& a, int elem_size, hipStream_t stream)
{
    const dim3 per_vertex((a.n + 255) / 256), threads(256);
    const int rows = a.r1 - a.r0;
    if (rows > 0) {
        const dim3 cols((unsigned)((a.r1 + COLS_PER_WG - 1) / COLS_PER_WG), (unsigned)a.n_chunks); // columns >= r1 - 1 have no row below them here
        const dim3 row_tiles((unsigned)((rows + ROWS_PER_WG - 1) / ROWS_PER_WG));
        const size_t lds = a.pow_in_lds ? (size_t)a.pow_n * sizeof(double) : 0;
#define MST_PASSES(T, K, P)                                                                         \
    hipLaunchKernelGGL((boruvka_row_kernel<T, K, P>), row_tiles, threads, lds, stream, a);           \
    hipLaunchKernelGGL((boruvka_col_kernel<T, K, P>), cols, threads, lds, stream, a);
#define MST_PASSES_T(T)                                                                             \
    if (a.kind != 1) { MST_PASSES(T, 0, false) }                                                    \
    else if (a.pow_in_lds) { MST_PASSES(T, 1, true) }                                               \
    else { MST_PASSES(T, 1, false) }
        if (elem_size == 2) { MST_PASSES_T(uint16_t) } else { MST_PASSES_T(uint32_t) }
#undef MST_PASSES_T
#undef MST_PASSES
    }
    hipLaunchKernelGGL(boruvka_fold_kernel, per_vertex, threads, 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_boruvka_merge(const BoruvkaArgs& a, const MstKey* gathered, int n_parts, hipStream_t stream)
{
    const dim3 per_vertex((a.n + 255) / 256), threads(256);
    hipLaunchKernelGGL(boruvka_reset_kernel, per_vertex, threads, 0, stream, a);
    hipLaunchKernelGGL(boruvka_gather_kernel, per_vertex, threads, 0, stream, a, gathered, n_parts);
    hipLaunchKernelGGL(boruvka_pick_kernel, per_vertex, threads, 0, stream, a);
    hipLaunchKernelGGL(boruvka_hook_kernel, per_vertex, threads, 0, stream, a);
    hipLaunchKernelGGL(boruvka_uncycle_kernel, per_vertex, threads, 0, stream, a);
    hipLaunchKernelGGL(boruvka_relabel_kernel, per_vertex, threads, 0, stream, a);
    return hipGetLastError();
}

} // namespace lcsgpu
