// clarans_kernels.hip -- CLARANS k-medoids of the MedoidTree heuristic on the device.
//
// What it reproduces: CLARANS::operator() (reference tree/Clustering.cpp:17-305) as called by
// FastTree::clusterSeeds (tree/FastTree.cpp:366-436) on the float distance triangle of the sample.
// The search is a chain of "steps": draw a non-medoid position xx, evaluate for every medoid slot
// k the cost change of replacing medoid k by candidate[xx], accept the best k if it lowers the
// cost, and stop after `corrected` steps without an accept.  The state only changes on an accept,
// and the positions xx come from a generator that does not look at the state, so ALL steps up to
// the next accept can be evaluated at once from the same state:
//   clarans_eval_kernel   one workgroup per pending step (+1 that keeps the running cost), lane = medoid
//                         slot k, every lane accumulates deltas[k] over the non-medoids in ascending
//                         position -- the reference's float additions in the reference's order;
//   clarans_apply_kernel  takes the FIRST step of the window whose best delta is negative, swaps, and
//                         re-derives nearest / second-nearest medoid of every non-medoid exactly as
//                         the reference's update branch does (one lane per non-medoid).
// One round = these two launches; the host enqueues rounds in batches and looks at the `done` flag
// between batches.  Ties, comparison directions and float operation order follow the reference
// line by line; the running cost is summed sequentially from a per-round log of its addends.
//
// Layout: D = float triangle over the sample members (D[i*(i-1)/2 + j], j < i); DM[y*k + mm] =
// distance of member y to the medoid in slot mm (kept in step with the swaps so that the reference's
// updateAssignment scan reads one contiguous row instead of k scattered triangle entries).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <climits>
#include <cstdint>

#include "lcs_kernels.h"

namespace lcsgpu {

namespace {

__device__ __forceinline__ size_t tri_at(int i, int j)
{
    return i >= j ? (size_t)j + (size_t)i * (i - 1) / 2 : (size_t)i + (size_t)j * (j - 1) / 2;
}

// CLARANS::updateAssignment (Clustering.cpp:262-305) over the member's DM row
__device__ __forceinline__ void scan_row(const float* __restrict__ row, int k, float& dn, float& ds, int& an, int& as)
{
    float bn = FLT_MAX, bs = FLT_MAX;
    int in = -1, is = -1;
    for (int mm = 0; mm < k; ++mm) {
        const float d = row[mm];
        if (d < bn) { bs = bn; is = in; bn = d; in = mm; }
        else if (d < bs) { bs = d; is = mm; }
    }
    dn = bn; ds = bs; an = in; as = is;
}

enum { ST_P = 0, ST_DONE = 1, ST_LOG_LEN = 2, ST_ROUNDS = 3, ST_ARRIVE = 4, ST_COST = 5, ST_ERR = 6 };

} // namespace

// float distances of the sample: D[tri(i,j)] = transform(LCS(ref = ids[i], partner = ids[j])), j < i
// (Transform<float, ...>, tree/AbstractTreeGenerator.hpp:28-82, the float table built at upload)
template <typename T>
__global__ __launch_bounds__(256) void subset_dist_kernel(const T* __restrict__ lcs, const int32_t* __restrict__ ids,
                                                          const uint32_t* __restrict__ lens,
                                                          const float* __restrict__ pow_f32, int kind,
                                                          float* __restrict__ D)
{
    const int i = blockIdx.x + 1;
    const uint32_t len_i = lens[ids[i]];
    const size_t row = (size_t)i * (i - 1) / 2;
    for (int j = threadIdx.x; j < i; j += 256) {
        const uint32_t l = lcs[row + j];
        const uint32_t indel = len_i + lens[ids[j]] - 2u * l;
        float d;
        if (l == 0) d = FLT_MAX;
        else if (kind == 1) d = __fdiv_rn(pow_f32[indel], (float)l);
        else d = __fdiv_rn((float)indel, (float)l);
        D[row + j] = d;
    }
}

// Start of one local search (Clustering.cpp:49-79): medoid bookkeeping, every non-medoid's DM row
// and assignment, the addends of the initial cost in position order.
__global__ __launch_bounds__(256) void clarans_init_kernel(ClaransArgs a)
{
    const int pos = blockIdx.x * 256 + threadIdx.x;
    const int k = a.n_medoids;
    if (pos == 0) {
        a.state[ST_DONE] = 0;
        a.state[ST_LOG_LEN] = a.n_elems - k;
        a.state[ST_ROUNDS] = 0;
        a.state[ST_ARRIVE] = 0;
        a.state[ST_COST] = __float_as_int(0.0f);
    }
    if (pos >= a.n_elems) return;
    const int y = a.cand[pos];
    if (pos < k) {
        a.dn[y] = 0.0f; a.ds[y] = -1.0f; a.an[y] = -1; a.as_[y] = -1;
        return;
    }
    float* row = a.DM + (size_t)y * k;
    for (int mm = 0; mm < k; ++mm) row[mm] = a.D[tri_at(a.cand[mm], y)];
    float dn, ds;
    int an, as;
    scan_row(row, k, dn, ds, an, as);
    a.dn[y] = dn; a.ds[y] = ds; a.an[y] = an; a.as_[y] = as;
    a.cost_log[pos - k] = dn;
}

// One workgroup per pending step b of the window [p, p + W): deltas[k] of candidate[draws[p + b]]
// (Clustering.cpp:93-118) and their first minimum over the free slots (cpp:121-122).
// Workgroup W adds the previous round's cost addends to the running cost, in order.
template <int KPT>
__global__ __launch_bounds__(128) void clarans_eval_kernel(ClaransArgs a, int W)
{
    constexpr int CH = 512;
    __shared__ float4 s_e[CH];
    __shared__ float s_v[128];
    __shared__ int s_k[128];
    if (a.state[ST_DONE]) return;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int k = a.n_medoids, n = a.n_elems;
    if (b == W) {
        const int len = a.state[ST_LOG_LEN];
        if (len == 0) return;
        float c = __int_as_float(a.state[ST_COST]);
        float* s_f = reinterpret_cast<float*>(s_e);
        for (int c0 = 0; c0 < len; c0 += 4 * CH) {
            const int cnt = min(4 * CH, len - c0);
            for (int t = tid; t < 4 * CH; t += 128) s_f[t] = t < cnt ? a.cost_log[c0 + t] : 0.0f;
            __syncthreads();
            if (tid == 0) {
                const int q4 = (cnt + 3) / 4; // the padding adds +0.0f: identity on a non-negative-zero sum
                for (int t = 0; t < q4; ++t) {
                    const float4 v = s_e[t];
                    c = __fadd_rn(c, v.x); c = __fadd_rn(c, v.y); c = __fadd_rn(c, v.z); c = __fadd_rn(c, v.w);
                }
            }
            __syncthreads();
        }
        if (tid == 0) a.state[ST_COST] = __float_as_int(c);
        return;
    }
    const int p = a.state[ST_P];
    if (p + W > a.draws_len) {
        if (tid == 0) a.state[ST_ERR] = 1; // the host did not provide enough draws; apply ends the search
        return;
    }
    const int xx = a.draws[p + b];
    const int x = a.cand[xx];
    float acc[KPT];
#pragma unroll
    for (int q = 0; q < KPT; ++q) acc[q] = 0.0f;
    for (int c0 = k; c0 < n; c0 += CH) {
        const int cnt = min(CH, n - c0);
        for (int t = tid; t < cnt; t += 128) {
            const int yy = c0 + t;
            float4 e = make_float4(0.0f, 0.0f, __int_as_float(-1), 0.0f); // yy == xx: contributes nothing
            if (yy != xx) {
                const int y = a.cand[yy];
                const float dxy = a.D[tri_at(x, y)];
                const float dn = a.dn[y], ds = a.ds[y];
                const float m = ds < dxy ? ds : dxy;                // std::min(dxy, ds)
                const float change = __fsub_rn(dxy, dn);
                e.x = __fsub_rn(m, dn);                              // goes to deltas[nearest(y)]
                e.y = change < 0.0f ? change : 0.0f;                 // goes to every other slot when negative
                e.z = __int_as_float(a.an[y]);
            }
            s_e[t] = e;
        }
        __syncthreads();
#pragma unroll 8
        for (int t = 0; t < cnt; ++t) {
            const float4 e = s_e[t];
            const int nn = __float_as_int(e.z);
#pragma unroll
            for (int q = 0; q < KPT; ++q) acc[q] = __fadd_rn(acc[q], (tid + 128 * q) == nn ? e.x : e.y);
        }
        __syncthreads();
    }
    // std::min_element over slots [n_fixed, k): smallest value, earliest slot among equals
    float best = 0.0f;
    int bk = INT_MAX;
#pragma unroll
    for (int q = 0; q < KPT; ++q) {
        const int kk = tid + 128 * q;
        if (kk >= a.n_fixed && kk < k && (bk == INT_MAX || acc[q] < best)) { best = acc[q]; bk = kk; }
    }
    s_v[tid] = best;
    s_k[tid] = bk;
    __syncthreads();
    for (int s = 64; s > 0; s >>= 1) {
        if (tid < s) {
            const float v2 = s_v[tid + s];
            const int k2 = s_k[tid + s];
            const int k1 = s_k[tid];
            if (k2 != INT_MAX && (k1 == INT_MAX || v2 < s_v[tid] || (v2 == s_v[tid] && k2 < k1))) {
                s_v[tid] = v2;
                s_k[tid] = k2;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        a.res_delta[b] = s_v[0];
        a.res_mm[b] = s_k[0];
    }
}

// Accept the first improving step of the window (Clustering.cpp:124-238) or finish the search.
// One lane per non-medoid position; the workgroup that arrives last commits the swap.
__global__ __launch_bounds__(256) void clarans_apply_kernel(ClaransArgs a, int W)
{
    __shared__ int s_w;
    __shared__ int s_last;
    __shared__ int s_med[CLARANS_MAX_MEDOIDS];
    int* st = a.state;
    if (st[ST_DONE]) return;
    const int tid = threadIdx.x;
    const int k = a.n_medoids, n = a.n_elems;
    if (tid == 0) s_w = INT_MAX;
    __syncthreads();
    if (!st[ST_ERR])
        for (int w = tid; w < W; w += 256)
            if (a.res_delta[w] < 0.0f) atomicMin(&s_w, w);
    __syncthreads();
    const int w = s_w;
    const int p = st[ST_P];
    if (w == INT_MAX) { // `corrected` steps without an accept: this local search is over
        if (blockIdx.x == 0 && tid == 0) {
            st[ST_P] = p + (st[ST_ERR] ? 0 : W);
            st[ST_LOG_LEN] = 0;
            st[ST_DONE] = 1;
        }
        return;
    }
    const int xx = a.draws[p + w];
    const int mm_new = a.res_mm[w];
    const int x = a.cand[xx];         // the new medoid
    const int m_old = a.cand[mm_new]; // the medoid it replaces, now at position xx
    for (int i = tid; i < k; i += 256) s_med[i] = i == mm_new ? x : a.cand[i];
    __syncthreads();
    const int yy = k + blockIdx.x * 256 + tid;
    if (yy < n) {
        const int y = yy == xx ? m_old : a.cand[yy];
        float* row = a.DM + (size_t)y * k;
        float addend = 0.0f;
        float dn, ds;
        int an, as;
        if (yy == xx) {
            for (int mm = 0; mm < k; ++mm) row[mm] = a.D[tri_at(s_med[mm], y)];
            scan_row(row, k, dn, ds, an, as);
            a.dn[y] = dn; a.ds[y] = ds; a.an[y] = an; a.as_[y] = as;
            addend = dn;
        } else {
            const float d_new = a.D[tri_at(x, y)];
            row[mm_new] = d_new;
            const float dn_y = a.dn[y];
            const int an_y = a.an[y];
            if (an_y == mm_new) { // its medoid is the one that left
                const float ds_y = a.ds[y];
                if (d_new < ds_y) {
                    a.dn[y] = d_new;
                    addend = __fsub_rn(d_new, dn_y);
                } else {
                    scan_row(row, k, dn, ds, an, as);
                    a.dn[y] = dn; a.ds[y] = ds; a.an[y] = an; a.as_[y] = as;
                    addend = __fsub_rn(ds_y, dn_y);
                }
            } else if (d_new < dn_y) {
                a.ds[y] = dn_y; a.as_[y] = an_y;
                a.dn[y] = d_new; a.an[y] = mm_new;
                addend = __fsub_rn(d_new, dn_y);
            } else {
                const float ds_y = a.ds[y];
                if (a.as_[y] != mm_new && d_new < ds_y) {
                    a.ds[y] = d_new; a.as_[y] = mm_new;
                } else {
                    scan_row(row, k, dn, ds, an, as);
                    a.dn[y] = dn; a.ds[y] = ds; a.an[y] = an; a.as_[y] = as;
                }
            }
        }
        a.cost_log[1 + yy - k] = addend;
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(&st[ST_ARRIVE], 1) == (int)gridDim.x - 1;
    __syncthreads();
    if (s_last && tid == 0) {
        __threadfence();
        a.cost_log[0] = -a.dn[x]; // cost -= dists_nearest[m_new], before the loop's addends
        a.dn[x] = 0.0f; a.ds[x] = -1.0f; a.an[x] = -1; a.as_[x] = -1;
        a.cand[mm_new] = x;
        a.cand[xx] = m_old;
        st[ST_P] = p + w + 1;
        st[ST_LOG_LEN] = 1 + n - k;
        st[ST_ROUNDS] = st[ST_ROUNDS] + 1;
        st[ST_ARRIVE] = 0;
    }
}

hipError_t launch_subset_distances(const void* lcs, int elem_size, const int32_t* ids, const uint32_t* lens,
                                   const float* pow_f32, int kind, int n, float* D, hipStream_t stream)
{
    if (n < 2) return hipSuccess;
    if (elem_size == 2)
        hipLaunchKernelGGL(subset_dist_kernel<uint16_t>, dim3(n - 1), dim3(256), 0, stream, (const uint16_t*)lcs, ids,
                           lens, pow_f32, kind, D);
    else
        hipLaunchKernelGGL(subset_dist_kernel<uint32_t>, dim3(n - 1), dim3(256), 0, stream, (const uint32_t*)lcs, ids,
                           lens, pow_f32, kind, D);
    return hipGetLastError();
}

hipError_t launch_clarans_init(const ClaransArgs& a, hipStream_t stream)
{
    hipLaunchKernelGGL(clarans_init_kernel, dim3((a.n_elems + 255) / 256), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// `rounds` x (evaluate a window, apply); the first window of a local search has `corrected` steps,
// the later ones corrected - 1 (the reference resets its step counter to 1 after an accept).
hipError_t launch_clarans_rounds(const ClaransArgs& a, int corrected, bool first_of_search, int rounds,
                                 hipStream_t stream)
{
    const int kpt = (a.n_medoids + 127) / 128;
    const int apply_blocks = (a.n_elems - a.n_medoids + 255) / 256;
    for (int r = 0; r < rounds; ++r) {
        const int W = (first_of_search && r == 0) ? corrected : (corrected > 0 ? corrected - 1 : 0);
        const dim3 grid(W + 1), block(128);
        if (kpt <= 1) hipLaunchKernelGGL(clarans_eval_kernel<1>, grid, block, 0, stream, a, W);
        else if (kpt <= 2) hipLaunchKernelGGL(clarans_eval_kernel<2>, grid, block, 0, stream, a, W);
        else if (kpt <= 4) hipLaunchKernelGGL(clarans_eval_kernel<4>, grid, block, 0, stream, a, W);
        else hipLaunchKernelGGL(clarans_eval_kernel<8>, grid, block, 0, stream, a, W);
        hipLaunchKernelGGL(clarans_apply_kernel, dim3(apply_blocks > 0 ? apply_blocks : 1), dim3(256), 0, stream, a, W);
    }
    return hipGetLastError();
}

} // namespace lcsgpu
