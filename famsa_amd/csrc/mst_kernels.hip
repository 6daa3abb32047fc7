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
// pow table + IEEE f64 division -- but only for candidates that can win: a float approximation of the
// quotient, exp2(0.75 log2(indel)) * rcp(l) on the transcendental unit (relative error < 2^-19: v_log_f32 /
// v_exp_f32 / v_rcp_f32 are 1-ulp instructions and |log2(indel)| <= 32), is compared first with a per-lane
// threshold kept just above the lane's current exact best (x (1 + 2^-14)), so a candidate that fails it is
// provably larger than the best and the table gather, the f64 division and the 128-bit compare are skipped.
// Measured at n = 100 000 (10 GB of u16 per pass): exact distance for every pair 7.2 / 9.8 ms (row / column
// pass, ALU-bound); pre-filter with a float pow TABLE 7.8 / 5.5 ms -- the per-lane gather of 64 random table
// entries costs the CU's single texture-address unit ~25 cycles per wave; no table: see DESIGN.md.
//
// Valid when d(u, v) does not depend on which endpoint is the ref: always for the triangle's own
// orientation (what SLINK sees), and for MSTPrim's orientation when no uploaded sequence is orientation
// sensitive (SURVEY note Q); otherwise the caller keeps the Prim kernel.
#include <hip/hip_runtime.h>

#include <cstdint>

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

// a lane's running best: exact key + the float threshold that admits every candidate able to beat it
struct Best {
    unsigned long long d = NO_D, id = NO_ID;
    float thr = __builtin_inff();
};

// numerator of the float approximation: indel^0.75 (KIND 1) or indel, without touching memory
template <int KIND>
__device__ __forceinline__ float approx_numerator(const BoruvkaArgs&, uint32_t indel)
{
    const float x = (float)indel;
    if (KIND != 1) return x;
    return __builtin_amdgcn_exp2f(0.75f * __builtin_amdgcn_logf(x)); // indel == 0: exp2(-inf) = 0
}

// candidate pair (lo < hi) with LCS l and indel = len_ref + len_partner - 2 l (ref = the larger id: the
// triangle's orientation); num = approx_numerator(indel)
template <int KIND>
__device__ __forceinline__ void consider(const BoruvkaArgs& a, Best& b, uint32_t l, uint32_t indel, float num, uint32_t lo,
                                         uint32_t hi)
{
    const float approx = num * __builtin_amdgcn_rcpf((float)l); // l == 0: inf or NaN -> never "greater" below
    if (approx > b.thr)
        return; // exact d >= approx / (1 + 2^-19) > best d: cannot win, not even a tie
    double d;
    if (l == 0) d = 1.7976931348623155e308; // nextafter(DBL_MAX, 0), hpp:61,73
    else if (KIND == 1) d = a.pow_table[indel] / (double)l;
    else d = (double)indel / (double)l;
    const unsigned long long db = (unsigned long long)__double_as_longlong(d);
    const unsigned long long id = ~(((unsigned long long)lo << 32) + hi);
    if (key_less(db, id, b.d, b.id)) {
        b.d = db;
        b.id = id;
        b.thr = __double2float_ru(d) * 1.00006104f; // x (1 + 2^-14), stays >= d (1 + 2^-15)
    }
}

} // namespace

__global__ __launch_bounds__(256) void boruvka_init_kernel(BoruvkaArgs a)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v < a.n) a.comp[v] = v;
    if (v == 0) a.counters[0] = 0; // edges recorded so far
}

// best edge of vertex v = r0 + blockIdx.x among u < v (row v of the triangle), to another component
template <typename T, int KIND>
__global__ __launch_bounds__(256) void boruvka_row_kernel(BoruvkaArgs a)
{
    __shared__ unsigned long long s_d[256], s_i[256];
    const int v = a.r0 + blockIdx.x, tid = threadIdx.x;
    const int cv = a.comp[v];
    const uint32_t len_v = a.lens[v];
    const T* row = (const T*)a.tri + ((int64_t)v * (v - 1) / 2 - a.off);
    Best b;
    for (int u0 = tid; u0 < v; u0 += 256 * 8) { // 8 independent loads per lane in flight
        uint32_t l[8], indel[8];
        int cu[8];
        float num[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int u = u0 + 256 * k;
            const bool in = u < v;
            l[k] = in ? (uint32_t)row[u] : 0u;
            cu[k] = in ? a.comp[u] : cv;
            indel[k] = in ? a.lens[u] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            indel[k] += len_v - 2u * l[k];
            num[k] = approx_numerator<KIND>(a, indel[k]);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (cu[k] != cv) consider<KIND>(a, b, l[k], indel[k], num[k], (uint32_t)(u0 + 256 * k), (uint32_t)v);
    }
    s_d[tid] = b.d;
    s_i[tid] = b.id;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s && key_less(s_d[tid + s], s_i[tid + s], s_d[tid], s_i[tid])) {
            s_d[tid] = s_d[tid + s];
            s_i[tid] = s_i[tid + s];
        }
        __syncthreads();
    }
    if (tid == 0) a.row_best[v] = MstKey{s_d[0], s_i[0]};
}

// best edge of vertex v (lane = column) among the rows u > v of one row chunk of the block -> part[chunk][v].
// The row index is uniform across the workgroup, so comp[u] / lens[u] are scalar loads.
template <typename T, int KIND>
__global__ __launch_bounds__(256) void boruvka_col_kernel(BoruvkaArgs a)
{
    const int c0 = blockIdx.x * 256;
    const int v = c0 + threadIdx.x;
    const int chunk = blockIdx.y;
    const int u0 = a.r0 + chunk * a.rows_per_chunk, u1 = min(a.r1, u0 + a.rows_per_chunk);
    if (u0 >= u1 || c0 + 1 >= u1) return; // empty chunk, or every column of this workgroup lies at or above its last row: fold skips it
    Best b;
    if (v < a.n) {
        const int cv = a.comp[v];
        const uint32_t len_v = a.lens[v];
        const T* col = (const T*)a.tri + ((int64_t)v - a.off);
        for (int ub = max(u0, c0 + 1); ub < u1; ub += 8) { // 8 independent loads per lane in flight
            uint32_t l[8], indel[8];
            int cu[8];
            float num[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int u = ub + k;
                const bool in = u < u1 && u > v;
                const int us = min(u, u1 - 1); // uniform, always a valid row
                cu[k] = a.comp[us];
                indel[k] = a.lens[us];
                l[k] = in ? (uint32_t)col[(int64_t)u * (u - 1) / 2] : 0u;
                if (!in) cu[k] = cv;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                indel[k] += len_v - 2u * l[k];
                num[k] = approx_numerator<KIND>(a, indel[k]);
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (cu[k] != cv) consider<KIND>(a, b, l[k], indel[k], num[k], (uint32_t)v, (uint32_t)(ub + k));
        }
        a.part[(size_t)chunk * a.n + v] = MstKey{b.d, b.id};
    }
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
    const int c0 = v & ~255;
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
        const dim3 cols((unsigned)((a.r1 + 255) / 256), (unsigned)a.n_chunks); // columns >= r1 - 1 have no row below them here
#define MST_PASSES(T, K)                                                                            \
    hipLaunchKernelGGL((boruvka_row_kernel<T, K>), dim3(rows), threads, 0, stream, a);               \
    hipLaunchKernelGGL((boruvka_col_kernel<T, K>), cols, threads, 0, stream, a);
        if (elem_size == 2) {
            if (a.kind == 1) { MST_PASSES(uint16_t, 1) } else { MST_PASSES(uint16_t, 0) }
        } else {
            if (a.kind == 1) { MST_PASSES(uint32_t, 1) } else { MST_PASSES(uint32_t, 0) }
        }
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
