// tree_kernels.hip -- device-side reducers over the resident LCS triangle.
//
// prim_step_kernel: one launch = one step of Prim's algorithm on the implicit complete graph,
// exactly the recurrence of MSTPrim::run_view (reference tree/MSTPrim.cpp:356-533):
//   key[v] = min(key[v], (d(cur, v), ~pack(min(cur,v), max(cur,v))))   lexicographic (double, u64)
//   next   = argmin over unprocessed v of key[v]
// The steps are strictly sequential; the only synchronisation is the kernel boundary: every
// workgroup of launch k first reduces the per-workgroup minima that launch k-1 left in global
// memory (a few KB) to learn `cur`, then relaxes its slice of the keys against `cur` and leaves
// its own minimum for launch k+1 (two partial buffers, alternating).  Workgroup 0 also records
// the chosen edge.  HBM-bound: per step n LCS values (2 B each; contiguous for v < cur, one
// 64-B sector each for v > cur) + 16 B of key per vertex.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lcs_kernels.h"

namespace lcsgpu {

__device__ __forceinline__ bool key_less(double d1, uint64_t i1, double d2, uint64_t i2)
{
    return d1 < d2 || (d1 == d2 && i1 < i2);
}

__device__ __forceinline__ uint64_t pack_ids(uint32_t a, uint32_t b) // ids_to_uint64, tree/MSTPrim.h:432-439
{
    return a < b ? ((uint64_t)a << 32) + b : ((uint64_t)b << 32) + a;
}

template <typename T>
__global__ __launch_bounds__(256) void prim_step_kernel(PrimArgs a, int step)
{
    __shared__ double s_d[256];
    __shared__ uint64_t s_i[256];
    __shared__ int s_v[256];
    const int tid = threadIdx.x;
    const int n = a.n;

    // ---- 1. the vertex chosen by the previous step (or vertex 0 at step 0) ----
    int cur = 0;
    if (step > 0) {
        const PrimPartial* prev = a.partials + (size_t)((step - 1) & 1) * a.n_blocks;
        double bd = 1.7976931348623157e308;
        uint64_t bi = ~0ull;
        int bv = -1;
        for (int b = tid; b < a.n_blocks; b += 256) {
            const PrimPartial p = prev[b];
            if (p.v >= 0 && (bv < 0 || key_less(p.d, p.id, bd, bi))) {
                bd = p.d;
                bi = p.id;
                bv = p.v;
            }
        }
        s_d[tid] = bd;
        s_i[tid] = bi;
        s_v[tid] = bv;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s && s_v[tid + s] >= 0 &&
                (s_v[tid] < 0 || key_less(s_d[tid + s], s_i[tid + s], s_d[tid], s_i[tid]))) {
                s_d[tid] = s_d[tid + s];
                s_i[tid] = s_i[tid + s];
                s_v[tid] = s_v[tid + s];
            }
            __syncthreads();
        }
        cur = s_v[0];
        if (blockIdx.x == 0 && tid == 0) { // edge number `step` joins cur to the tree
            const uint64_t packed = ~s_i[0];
            a.edges[step - 1].from = (int32_t)(packed >> 32);
            a.edges[step - 1].to = (int32_t)(packed & 0xffffffffull);
            a.edges[step - 1].dist = s_d[0];
        }
        __syncthreads();
    }
    if (step >= n - 1) // the finalising launch only records the last edge
        return;

    // ---- 2. relax my vertices against cur, keep my minimum ----
    const int v = blockIdx.x * 256 + tid;
    double md = 1.7976931348623157e308;
    uint64_t mi = ~0ull;
    int mv = -1;
    if (v < n) {
        if (v == cur) a.processed[v] = 1;
        if (v != cur && !a.processed[v]) {
            // LCS(ref = cur, partner = v): the triangle holds (ref = larger id, partner = smaller id); the
            // two orientations differ only if the ref is orientation sensitive -> side tables
            uint32_t l;
            const int qc = a.qindex ? a.qindex[cur] : -1;
            const int qv = a.qindex ? a.qindex[v] : -1;
            if (qc >= 0)
                l = a.q_rows[(size_t)qc * n + v]; // ref = cur is sensitive: its own row
            else if (qv >= 0 && v > cur)
                l = a.q_cols[(size_t)cur * a.n_q + qv]; // triangle would use ref = v (sensitive)
            else {
                const uint64_t hi = v > cur ? v : cur, lo = v > cur ? cur : v;
                l = ((const T*)a.tri)[hi * (hi - 1) / 2 + lo];
            }
            const uint32_t indel = a.lens[cur] + a.lens[v] - 2u * l;
            double d;
            if (l == 0)
                d = 1.7976931348623155e308; // nextafter(DBL_MAX, 0)
            else if (a.kind == 1)
                d = a.pow_table[indel] / (double)l;
            else
                d = (double)indel / (double)l;
            double kd = a.key_d[v];
            uint64_t ki = a.key_id[v];
            if (d <= kd) {
                const uint64_t id = ~pack_ids((uint32_t)cur, (uint32_t)v);
                if (key_less(d, id, kd, ki)) {
                    kd = d;
                    ki = id;
                    a.key_d[v] = kd;
                    a.key_id[v] = ki;
                }
            }
            md = kd;
            mi = ki;
            mv = v;
        }
    }
    s_d[tid] = md;
    s_i[tid] = mi;
    s_v[tid] = mv;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s && s_v[tid + s] >= 0 &&
            (s_v[tid] < 0 || key_less(s_d[tid + s], s_i[tid + s], s_d[tid], s_i[tid]))) {
            s_d[tid] = s_d[tid + s];
            s_i[tid] = s_i[tid + s];
            s_v[tid] = s_v[tid + s];
        }
        __syncthreads();
    }
    if (tid == 0) {
        PrimPartial* mine = a.partials + (size_t)(step & 1) * a.n_blocks + blockIdx.x;
        mine->d = s_d[0];
        mine->id = s_i[0];
        mine->v = s_v[0];
    }
}

__global__ void prim_init_kernel(PrimArgs a)
{
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v < a.n) {
        a.key_d[v] = 1.7976931348623157e308; // numeric_limits<double>::max(), MSTPrim.cpp:289
        a.key_id[v] = 0;
        a.processed[v] = 0;
    }
}

hipError_t launch_prim(const PrimArgs& a, int elem_size, hipStream_t stream)
{
    const int blocks = a.n_blocks;
    hipLaunchKernelGGL(prim_init_kernel, dim3(blocks), dim3(256), 0, stream, a);
    for (int step = 0; step < a.n; ++step) { // n-1 relaxing launches + 1 finalising launch
        if (elem_size == 2)
            hipLaunchKernelGGL(prim_step_kernel<uint16_t>, dim3(step >= a.n - 1 ? 1 : blocks), dim3(256), 0, stream, a, step);
        else
            hipLaunchKernelGGL(prim_step_kernel<uint32_t>, dim3(step >= a.n - 1 ? 1 : blocks), dim3(256), 0, stream, a, step);
    }
    return hipGetLastError();
}

} // namespace lcsgpu
