// mst_kernels.hip -- the minimum spanning tree of the uploaded set by Boruvka rounds over the LCS
// triangle in HBM.
//
// MSTPrim (reference tree/MSTPrim.cpp:356-533) orders edges by the strict total order
// (d, ~pack(min id, max id)) -- smaller distance first, then the larger packed id -- so the MST is
// unique and any exact MST algorithm yields the reference's edge set; Prim's insertion order from
// vertex 0 is then a walk over those n-1 edges (done by the caller).  Prim itself needs n-1
// dependent steps (one launch each, ~7 us: 0.7 s at n = 100 000); Boruvka needs <= log2(n) rounds of
// streaming passes over the triangle (2 B per pair, HBM-bound: ~2.5 ms per round at n = 100 000):
//   1. every vertex's best edge to another component: row part (u < v: its own contiguous row) and
//      column part (u > v: lanes = consecutive columns, walking down the rows -- coalesced);
//   2. every component's best edge (two 64-bit atomic-min phases: distance bits, then id);
//   3. hook each component to the other end of its edge (mutual choices: the smaller root stays a
//      root and the edge is recorded once), then relabel the vertices.
// Valid when d(u, v) does not depend on which endpoint is the ref: always for the triangle's own
// orientation (what SLINK sees), and for MSTPrim's orientation when no uploaded sequence is
// orientation sensitive (SURVEY note Q); otherwise the caller keeps the Prim kernel.
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

template <typename T>
__device__ __forceinline__ unsigned long long dist_bits(const BoruvkaArgs& a, uint32_t l, uint32_t len1, uint32_t len2)
{
    const uint32_t indel = len1 + len2 - 2u * l;
    double d;
    if (l == 0) d = 1.7976931348623155e308; // nextafter(DBL_MAX, 0)
    else if (a.kind == 1) d = a.pow_table[indel] / (double)l;
    else d = (double)indel / (double)l;
    return (unsigned long long)__double_as_longlong(d);
}

} // namespace

__global__ __launch_bounds__(256) void boruvka_init_kernel(BoruvkaArgs a)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v < a.n) a.comp[v] = v;
    if (v == 0) a.counters[0] = 0; // edges recorded so far
}

// best edge of vertex v = blockIdx.x among u < v (row v of the triangle), to another component
template <typename T>
__global__ __launch_bounds__(256) void boruvka_row_kernel(BoruvkaArgs a)
{
    __shared__ unsigned long long s_d[256], s_i[256];
    const int v = blockIdx.x, tid = threadIdx.x;
    const int cv = a.comp[v];
    const uint32_t len_v = a.lens[v];
    const T* row = (const T*)a.tri + (size_t)v * (v > 0 ? v - 1 : 0) / 2;
    unsigned long long bd = NO_D, bi = NO_ID;
    for (int u0 = tid; u0 < v; u0 += 256 * 8) { // 8 independent loads per lane in flight
        uint32_t l[8], len_u[8];
        int cu[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int u = u0 + 256 * k;
            const bool in = u < v;
            l[k] = in ? (uint32_t)row[u] : 0u;
            cu[k] = in ? a.comp[u] : cv;
            len_u[k] = in ? a.lens[u] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (cu[k] == cv) continue;
            const int u = u0 + 256 * k;
            const unsigned long long d = dist_bits<T>(a, l[k], len_v, len_u[k]);
            const unsigned long long id = ~pack_ids((uint32_t)u, (uint32_t)v);
            if (key_less(d, id, bd, bi)) { bd = d; bi = id; }
        }
    }
    s_d[tid] = bd;
    s_i[tid] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s && key_less(s_d[tid + s], s_i[tid + s], s_d[tid], s_i[tid])) {
            s_d[tid] = s_d[tid + s];
            s_i[tid] = s_i[tid + s];
        }
        __syncthreads();
    }
    if (tid == 0) {
        a.best_d[v] = s_d[0];
        a.best_id[v] = s_i[0];
    }
}

// best edge of vertex v (lane = column) among the rows u > v of one row chunk -> partial[chunk][v]
template <typename T>
__global__ __launch_bounds__(256) void boruvka_col_kernel(BoruvkaArgs a)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    const int chunk = blockIdx.y;
    const int u0 = chunk * a.rows_per_chunk, u1 = min(a.n, u0 + a.rows_per_chunk);
    unsigned long long bd = NO_D, bi = NO_ID;
    if (v < a.n) {
        const int cv = a.comp[v];
        const uint32_t len_v = a.lens[v];
        for (int ub = max(u0, v + 1); ub < u1; ub += 8) { // 8 independent loads per lane in flight
            uint32_t l[8], len_u[8];
            int cu[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int u = ub + k;
                const bool in = u < u1;
                l[k] = in ? (uint32_t)((const T*)a.tri)[(size_t)u * (u - 1) / 2 + v] : 0u;
                cu[k] = in ? a.comp[u] : cv;
                len_u[k] = in ? a.lens[u] : 0u;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (cu[k] == cv) continue;
                const int u = ub + k;
                const unsigned long long d = dist_bits<T>(a, l[k], len_u[k], len_v);
                const unsigned long long id = ~pack_ids((uint32_t)v, (uint32_t)u);
                if (key_less(d, id, bd, bi)) { bd = d; bi = id; }
            }
        }
        a.part_d[(size_t)chunk * a.n + v] = bd;
        a.part_id[(size_t)chunk * a.n + v] = bi;
    }
}

// fold the column partials into best[v]; first atomic phase of the per-component minimum
__global__ __launch_bounds__(256) void boruvka_fold_kernel(BoruvkaArgs a)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= a.n) return;
    unsigned long long bd = a.best_d[v], bi = a.best_id[v];
    for (int c = 0; c < a.n_chunks; ++c) {
        const unsigned long long d = a.part_d[(size_t)c * a.n + v], id = a.part_id[(size_t)c * a.n + v];
        if (key_less(d, id, bd, bi)) { bd = d; bi = id; }
    }
    a.best_d[v] = bd;
    a.best_id[v] = bi;
    if (bi != NO_ID) atomicMin(&a.cb_d[a.comp[v]], bd);
}

__global__ __launch_bounds__(256) void boruvka_reset_kernel(BoruvkaArgs a)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= a.n) return;
    a.cb_d[v] = NO_D;
    a.cb_id[v] = NO_ID;
    a.parent[v] = v;
}

// second atomic phase: among the vertices that reach their component's smallest distance, the smallest id
__global__ __launch_bounds__(256) void boruvka_pick_kernel(BoruvkaArgs a)
{
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= a.n) return;
    const unsigned long long bd = a.best_d[v], bi = a.best_id[v];
    if (bi == NO_ID) return;
    const int c = a.comp[v];
    if (bd == a.cb_d[c]) atomicMin(&a.cb_id[c], bi);
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

// one round; the caller swaps comp / comp_next afterwards and reads counters[0]
hipError_t launch_boruvka_round(const BoruvkaArgs& a, int elem_size, hipStream_t stream)
{
    const dim3 per_vertex((a.n + 255) / 256), threads(256);
    hipLaunchKernelGGL(boruvka_reset_kernel, per_vertex, threads, 0, stream, a);
    if (elem_size == 2) {
        hipLaunchKernelGGL(boruvka_row_kernel<uint16_t>, dim3(a.n), threads, 0, stream, a);
        hipLaunchKernelGGL(boruvka_col_kernel<uint16_t>, dim3(per_vertex.x, a.n_chunks), threads, 0, stream, a);
    } else {
        hipLaunchKernelGGL(boruvka_row_kernel<uint32_t>, dim3(a.n), threads, 0, stream, a);
        hipLaunchKernelGGL(boruvka_col_kernel<uint32_t>, dim3(per_vertex.x, a.n_chunks), threads, 0, stream, a);
    }
    hipLaunchKernelGGL(boruvka_fold_kernel, per_vertex, threads, 0, stream, a);
    hipLaunchKernelGGL(boruvka_pick_kernel, per_vertex, threads, 0, stream, a);
    hipLaunchKernelGGL(boruvka_hook_kernel, per_vertex, threads, 0, stream, a);
    hipLaunchKernelGGL(boruvka_uncycle_kernel, per_vertex, threads, 0, stream, a);
    hipLaunchKernelGGL(boruvka_relabel_kernel, per_vertex, threads, 0, stream, a);
    return hipGetLastError();
}

} // namespace lcsgpu
