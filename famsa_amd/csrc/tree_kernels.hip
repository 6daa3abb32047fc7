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
#include "fasttree_kernels.h"

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

// =============================================================================================
// Device-side UPGMA over the resident triangle: the nearest-neighbour-array formulation the
// reference took from MUSCLE (tree/UPGMA.cpp:114-295), kept operation for operation because its
// tie rules (strict '<' in ascending scans => the smallest index wins) and its deliberately stale
// row minima decide the topology.  n-1 sequential merges, one launch per merge (upgma_step_kernel):
// apply the previous merge's row statistics, pick Lmin = argmin_j min_dist[j] (first minimum) and
// Rmin = nearest[Lmin], then D[Lmin,j] = average(D[Lmin,j], D[Rmin,j]) for every active j with the
// nearest-pointer rename and the per-workgroup minimum of the new row.
// Distances are float, produced on the device from the LCS triangle with the reference's
// Transform<float,...>: a host-built (float)pow(indel,0.75) table and IEEE float division.
// =============================================================================================
#include "dpp_min.h"

namespace lcsgpu {

static constexpr uint32_t UPGMA_NONE = 0x7FFFFFFFu;
static constexpr float UPGMA_BIG = 1e29f; // UPGMA::BIG_DIST, reference tree/UPGMA.h:106

__device__ __forceinline__ size_t tri_index(uint64_t i, uint64_t j) // TriangleMatrix::access
{
    return i >= j ? j + i * (i - 1) / 2 : i + j * (j - 1) / 2;
}

template <typename T>
__global__ __launch_bounds__(256) void upgma_dist_kernel(const T* __restrict__ lcs, const uint32_t* __restrict__ lens,
                                                         const float* __restrict__ pow_f32, int kind, int row0, size_t lcs_off,
                                                         float* __restrict__ D)
{
    // rows row0 .. of the triangle; lcs holds them from element lcs_off of the packed triangle on
    const int i = row0 + blockIdx.x; // row (>= 1)
    const uint32_t len_i = lens[i];
    const size_t row = (size_t)i * (i - 1) / 2;
    for (int j = threadIdx.x; j < i; j += 256) {
        const uint32_t l = lcs[row + j - lcs_off];
        const uint32_t indel = len_i + lens[j] - 2u * l;
        float d;
        if (l == 0)
            d = 3.40282347e38f; // (float) nextafter((double) FLT_MAX, 0) rounds back to FLT_MAX
        else if (kind == 1)
            d = __fdiv_rn(pow_f32[indel], (float)l);
        else
            d = __fdiv_rn((float)indel, (float)l);
        D[row + j] = d;
    }
}

// The same distances as a full symmetric matrix D[i*n + j] = D[j*n + i]: a merge then reads two contiguous rows
// (in the packed triangle the part j > Lmin of "row" Lmin is a column walk, one 32-byte sector per element, on the
// merge's dependent path).  One workgroup per 32 x 32 tile at or below the diagonal: the tile is written as it
// is read (rows i, coalesced along j) and, through LDS, transposed (rows j, coalesced along i).
template <typename T>
__global__ __launch_bounds__(256) void upgma_dist_square_kernel(const T* __restrict__ lcs, const uint32_t* __restrict__ lens,
                                                                const float* __restrict__ pow_f32, int kind, int n,
                                                                size_t ld, float* __restrict__ D, long long tile0, size_t lcs_off)
{
    __shared__ float tile[32][33];
    // tile (ti, tj), tj <= ti, from the linear workgroup id; a launch covers the tiles tile0 .. of whole tile rows, and
    // lcs holds those rows' values from element lcs_off of the packed triangle on
    const long long b = tile0 + blockIdx.x;
    int ti = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
    while ((long long)(ti + 1) * (ti + 2) / 2 <= b) ++ti;
    while ((long long)ti * (ti + 1) / 2 > b) --ti;
    const int tj = (int)(b - (long long)ti * (ti + 1) / 2);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5; // ty = 0..7
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = ti * 32 + ty + 8 * r, j = tj * 32 + tx;
        float d = 0.0f;
        if (i < n && j < i) {
            const uint32_t l = lcs[(size_t)i * (i - 1) / 2 + j - lcs_off];
            const uint32_t indel = lens[i] + lens[j] - 2u * l;
            if (l == 0)
                d = 3.40282347e38f; // (float) nextafter((double) FLT_MAX, 0) rounds back to FLT_MAX
            else if (kind == 1)
                d = __fdiv_rn(pow_f32[indel], (float)l);
            else
                d = __fdiv_rn((float)indel, (float)l);
            D[(size_t)i * ld + j] = d;
        }
        tile[ty + 8 * r][tx] = d;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int j = tj * 32 + ty + 8 * r, i = ti * 32 + tx; // element (j, i) of the upper half = tile[i][j]
        if (i < n && j < i) D[(size_t)j * ld + i] = tile[tx][ty + 8 * r];
    }
    if (ti == tj && threadIdx.x < 32) { // the diagonal is never read; keep it defined
        const int i = ti * 32 + threadIdx.x;
        if (i < n) D[(size_t)i * ld + i] = 0.0f;
    }
}

template <bool SQUARE>
__device__ __forceinline__ size_t upgma_index(const UpgmaArgs& a, uint64_t row, uint64_t col)
{
    return SQUARE ? (size_t)(row * (uint64_t)a.ld + col) : tri_index(row, col);
}

// initial row minima over the FULL row of x (columns y != x), first strict minimum in ascending y
template <bool SQUARE>
__global__ __launch_bounds__(256) void upgma_init_kernel(UpgmaArgs a)
{
    __shared__ float s_d[256];
    __shared__ uint32_t s_j[256];
    const int x = blockIdx.x, tid = threadIdx.x;
    float best = UPGMA_BIG;
    uint32_t bj = UPGMA_NONE;
    for (int y = tid; y < a.n; y += 256) {
        if (y == x) continue;
        const float d = a.D[upgma_index<SQUARE>(a, x, y)];
        if (d < best) { best = d; bj = y; } // ascending y within the thread
    }
    s_d[tid] = best;
    s_j[tid] = bj;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            const float d2 = s_d[tid + s];
            const uint32_t j2 = s_j[tid + s];
            if (d2 < s_d[tid] || (d2 == s_d[tid] && j2 < s_j[tid])) { s_d[tid] = d2; s_j[tid] = j2; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        a.min_dist[x] = s_d[0];
        a.nearest[x] = s_j[0];
        a.node_index[x] = x;
    }
}

// (value, index) minimum of a workgroup: smaller value, then smaller index; values >= UPGMA_BIG never win
// against the (UPGMA_BIG, UPGMA_NONE) start ("dtDist < dtMinDist" from BIG_DIST in the reference)
__device__ __forceinline__ void block_first_min(float& d, uint32_t& j, float* s_d, uint32_t* s_j)
{
    // inside a wave by DPP (wave_first_min), then the 4 wave results through LDS: two barriers per reduction
    wave_first_min(d, j);
    const int tid = threadIdx.x;
    __syncthreads(); // the previous reduction's readers are done with s_d / s_j
    if ((tid & 63) == 0) {
        s_d[tid >> 6] = d;
        s_j[tid >> 6] = j;
    }
    __syncthreads();
    d = s_d[0];
    j = s_j[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        const float d2 = s_d[w];
        const uint32_t j2 = s_j[w];
        if (d2 < d || (d2 == d && j2 < j)) { d = d2; j = j2; }
    }
}

__device__ __forceinline__ void take_first_min(float d, uint32_t j, float& bd, uint32_t& bj)
{
    if (d < UPGMA_BIG && (d < bd || (d == bd && j < bj))) { bd = d; bj = j; }
}

// first minimum of min_dist over the 256 rows of every workgroup (and that row's nearest): the global pick then only
// looks at these
__global__ __launch_bounds__(256) void upgma_block_min_kernel(UpgmaArgs a)
{
    __shared__ float s_d[256];
    __shared__ uint32_t s_j[256];
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    float d = UPGMA_BIG;
    uint32_t bj = UPGMA_NONE;
    if (j < (uint32_t)a.n) take_first_min(a.min_dist[j], j, d, bj);
    block_first_min(d, bj, s_d, s_j);
    if (threadIdx.x == 0) {
        a.bm_d[blockIdx.x] = d;
        a.bm_j[blockIdx.x] = bj;
        a.bm_near[blockIdx.x] = bj != UPGMA_NONE ? a.nearest[bj] : UPGMA_NONE;
    }
}

// (value, index, nearest-of-index) candidates of the pick
__device__ __forceinline__ void take_first_min3(float d, uint32_t j, uint32_t nr, float& bd, uint32_t& bj, uint32_t& bn)
{
    if (d < UPGMA_BIG && (d < bd || (d == bd && j < bj))) { bd = d; bj = j; bn = nr; }
}
// block_first_min that also hands every thread the third field of the winner
__device__ __forceinline__ void block_first_min3(float& d, uint32_t& j, uint32_t& nr, float* s_d, uint32_t* s_j, uint32_t* s_n)
{
    const uint32_t mine_j = j, mine_n = nr;
    block_first_min(d, j, s_d, s_j);
    // (a barrier lies between every reader of *s_n below and the next call's write: block_first_min starts with one)
    if (j != UPGMA_NONE && mine_j == j) *s_n = mine_n; // all holders of the winner hold the same record
    __syncthreads();
    nr = j != UPGMA_NONE ? *s_n : UPGMA_NONE;
}

// words of UpgmaArgs::sel: [4p + 0..1] = (Lmin, Rmin) of the merges of parity p; [8] = error flag;
// [16 + 4(2p + w) + 0..2] = (min_dist bits, row, its nearest) of the first minimum over the rows of the workgroup
// that owns Lmin (w = 0) / Rmin (w = 1) of the merge of parity p, WITHOUT the rows Lmin and Rmin themselves
constexpr int UPGMA_SEL_EXCL = 16;

// One launch = one merge of UPGMA::computeTree (tree/UPGMA.cpp:198-288).  The merges are strictly
// sequential and the only synchronisation is the kernel boundary, so every workgroup redoes the small
// serial part itself instead of waiting for a single-workgroup kernel:
//   1. finish merge it-1: the new cluster's row minimum = first minimum of the per-workgroup minima the
//      previous launch left (two buffers, alternating);
//   2. pick this merge: first minimum of min_dist over the active rows = first minimum of the
//      per-workgroup minima bm[], where the two workgroups that own the rows touched by merge it-1 are
//      replaced by the minima over their other rows which their owners left in sel[] (computed while merge it-1
//      was applied), completed by the merged row itself;
//   3. update its own 256 rows (new distances to the merged cluster, nearest-pointer rename), and, as the owner of
//      Lmin's / Rmin's rows, leave the minimum over the other rows for the next launch.
// A kernel boundary leaves nothing in the caches, so what a launch costs is its chain of DEPENDENT loads (~1.5-2 us
// each): every address of steps 1-2 is known before the launch (one level: bm[] carries each minimum's nearest, so
// Rmin needs no look-up), the only dependent level is the two distance rows of step 3.  (Round 1's form had four:
// the selection, the touched workgroups' rows, nearest[Lmin], the distance rows -- 11 us per merge.)
// Workgroup 0 does the bookkeeping writes of step 1; the owners write their bm[] entries; everything a
// launch writes that the same launch reads elsewhere is either unused there or overridden by the same patch, so
// the order in which workgroups run does not matter.
template <bool MODIFIED, bool SQUARE>
__global__ __launch_bounds__(256) void upgma_step_kernel(UpgmaArgs a, int it)
{
    __shared__ float s_d[256];
    __shared__ uint32_t s_j[256];
    __shared__ uint32_t s_n;
    const int tid = threadIdx.x, b = blockIdx.x, nb = a.n_blocks, n = a.n;
    const uint32_t j = (uint32_t)b * 256 + tid;
    const bool in = j < (uint32_t)n;
    const int pp = (it - 1) & 1, pc = it & 1;
    // ---- the one level of loads whose addresses are known before the launch ----
    const uint32_t my_node = in ? a.node_index[j] : UPGMA_NONE;
    const uint32_t my_near = in ? a.nearest[j] : UPGMA_NONE;
    const float my_min = in ? a.min_dist[j] : UPGMA_BIG;
    const uint32_t own_bm_j = a.bm_j[b];
    uint32_t Lp = UPGMA_NONE, Rp = UPGMA_NONE;
    float eLd = UPGMA_BIG, eRd = UPGMA_BIG;
    uint32_t eLj = UPGMA_NONE, eRj = UPGMA_NONE, eLn = UPGMA_NONE, eRn = UPGMA_NONE;
    float new_d = UPGMA_BIG;
    uint32_t new_j = UPGMA_NONE;
    if (it > 0) {
        Lp = a.sel[4 * pp + 0];
        Rp = a.sel[4 * pp + 1];
        const uint32_t* e = a.sel + UPGMA_SEL_EXCL + 8 * pp;
        eLd = __uint_as_float(e[0]); eLj = e[1]; eLn = e[2];
        eRd = __uint_as_float(e[4]); eRj = e[5]; eRn = e[6];
        const float* pd = a.part_d + (size_t)pp * nb;
        const uint32_t* pj = a.part_j + (size_t)pp * nb;
        for (int x = tid; x < nb; x += 256) take_first_min(pd[x], pj[x], new_d, new_j);
    }
    float cd = UPGMA_BIG;
    uint32_t cj = UPGMA_NONE, cn = UPGMA_NONE;
    {
        const uint32_t bLp = Lp >> 8, bRp = Rp >> 8; // UPGMA_NONE >> 8 is no workgroup
        for (int x = tid; x < nb; x += 256) {
            const float d = a.bm_d[x];
            const uint32_t dj = a.bm_j[x], dn = a.bm_near[x]; // loaded whatever x is: no load waits for Lp
            if ((uint32_t)x != bLp && (uint32_t)x != bRp) take_first_min3(d, dj, dn, cd, cj, cn);
        }
    }
    uint32_t nodeLp = UPGMA_NONE, nodeRp = UPGMA_NONE;
    const bool prev_ok = it > 0 && Lp != UPGMA_NONE && Rp != UPGMA_NONE; // (no previous pick: the input is degenerate, sel[8] says so)
    if (prev_ok && b == 0 && tid == 0) { // in flight during the reductions below
        nodeLp = a.node_index[Lp];
        nodeRp = a.node_index[Rp];
    }
    // ---- 1. finish merge it-1 ----
    if (it > 0) block_first_min(new_d, new_j, s_d, s_j);
    // ---- 2. pick ----
    const uint32_t bLp = Lp >> 8, bRp = Rp >> 8;
    float tLd = UPGMA_BIG, tRd = UPGMA_BIG; // the minima of the two touched workgroups as they are now
    uint32_t tLj = UPGMA_NONE, tRj = UPGMA_NONE, tLn = UPGMA_NONE, tRn = UPGMA_NONE;
    if (it > 0) {
        take_first_min3(eLd, eLj, eLn, tLd, tLj, tLn);
        take_first_min3(new_d, Lp, new_j, tLd, tLj, tLn); // the merged row: its minimum and nearest are still in flight
        if (bRp != bLp) take_first_min3(eRd, eRj, eRn, tRd, tRj, tRn);
        take_first_min3(tLd, tLj, tLn, cd, cj, cn);
        take_first_min3(tRd, tRj, tRn, cd, cj, cn);
    }
    block_first_min3(cd, cj, cn, s_d, s_j, &s_n);
    const uint32_t L = cj;
    uint32_t R = UPGMA_NONE;
    if (it < n - 1 && L != UPGMA_NONE) R = cn;
    if (b == 0 && tid == 0) {
        if (prev_ok) {
            a.left[it - 1] = (int32_t)nodeLp;
            a.right[it - 1] = (int32_t)nodeRp;
            a.node_index[Lp] = (uint32_t)n + (uint32_t)(it - 1);
            a.node_index[Rp] = UPGMA_NONE;
            a.min_dist[Lp] = new_d;
        }
        if (it < n - 1) {
            a.sel[4 * pc + 0] = L;
            a.sel[4 * pc + 1] = R;
            if (L == UPGMA_NONE || R == UPGMA_NONE) a.sel[8] = 1; // degenerate input (reference: UB)
        }
    }
    const bool touched = it > 0 && ((uint32_t)b == bLp || (uint32_t)b == bRp);
    if (touched && tid == 0) { // this workgroup's minimum changed with merge it-1
        const bool isL = (uint32_t)b == bLp;
        uint32_t nr = isL ? tLn : tRn;
        if (nr == R && R != UPGMA_NONE) nr = L; // ... and this merge renames
        a.bm_d[b] = isL ? tLd : tRd;
        a.bm_j[b] = isL ? tLj : tRj;
        a.bm_near[b] = nr;
    }
    if (it >= n - 1) return; // the last launch only finishes merge n-2
    // ---- 3. update my rows ----
    float nd = UPGMA_BIG;
    uint32_t nj = UPGMA_NONE;
    const bool row_ok = L != UPGMA_NONE && R != UPGMA_NONE && in && my_node != UPGMA_NONE && j != Rp && j != L && j != R;
    uint32_t near_j = UPGMA_NONE;
    if (row_ok) {
        const size_t vL = upgma_index<SQUARE>(a, L, j), vR = upgma_index<SQUARE>(a, R, j);
        const float dL = a.D[vL], dR = a.D[vR];
        float v;
        if (MODIFIED) // 0.05f * (x + y) + 0.9f * min(x, y), no contraction (reference UPGMA.cpp:32-34)
            v = __fadd_rn(__fmul_rn(0.05f, __fadd_rn(dL, dR)), __fmul_rn(0.9f, fminf(dL, dR)));
        else
            v = __fmul_rn(__fadd_rn(dL, dR), 0.5f);
        near_j = j == Lp ? new_j : my_near; // the row of the previous merge: its nearest is still in flight
        if (near_j == R) near_j = L;
        if (near_j != my_near) {
            a.nearest[j] = near_j;
            if (!touched && j == own_bm_j) a.bm_near[b] = near_j;
        }
        a.D[vL] = v;
        if (SQUARE) a.D[(size_t)j * (size_t)a.ld + L] = v; // the mirror: a strided store, off the dependent path
        nd = v;
        nj = j;
    }
    block_first_min(nd, nj, s_d, s_j);
    if (tid == 0) {
        a.part_d[(size_t)pc * nb + b] = nd;
        a.part_j[(size_t)pc * nb + b] = nj;
    }
    // the owners of Lmin's and Rmin's rows: the minimum over their OTHER rows, for the next launch's pick
    const uint32_t bL = L >> 8, bR = R >> 8;
    if (L != UPGMA_NONE && R != UPGMA_NONE && ((uint32_t)b == bL || (uint32_t)b == bR)) {
        float ed = UPGMA_BIG;
        uint32_t ej = UPGMA_NONE, en = UPGMA_NONE;
        if (row_ok) take_first_min3(j == Lp ? new_d : my_min, j, near_j, ed, ej, en);
        block_first_min3(ed, ej, en, s_d, s_j, &s_n);
        if (tid == 0) {
            uint32_t* e = a.sel + UPGMA_SEL_EXCL + 8 * pc + ((uint32_t)b == bL ? 0 : 4);
            e[0] = __float_as_uint(ed);
            e[1] = ej;
            e[2] = en;
        }
    }
}

// float distances of the rows [r0, r1) of the triangle (r0 a multiple of 32 in the square layout: whole tile rows);
// lcs = those rows' LCS values, i.e. the packed triangle from element r0 (r0 - 1) / 2 on
hipError_t launch_upgma_distances(const UpgmaArgs& a, const void* lcs, int elem_size, const uint32_t* lens, const float* pow_f32,
                                  int kind, int r0, int r1, hipStream_t stream)
{
    const int n = a.n;
    const size_t off = (size_t)r0 * (size_t)(r0 > 0 ? r0 - 1 : 0) / 2;
    if (r1 <= r0) return hipSuccess;
    if (a.square) {
        const long long t0 = r0 / 32, t1 = ((long long)r1 + 31) / 32, tile0 = t0 * (t0 + 1) / 2, tiles = t1 * (t1 + 1) / 2 - tile0;
        if (elem_size == 2)
            hipLaunchKernelGGL(upgma_dist_square_kernel<uint16_t>, dim3((unsigned)tiles), dim3(256), 0, stream, (const uint16_t*)lcs,
                               lens, pow_f32, kind, std::min(n, r1), (size_t)a.ld, a.D, tile0, off);
        else
            hipLaunchKernelGGL(upgma_dist_square_kernel<uint32_t>, dim3((unsigned)tiles), dim3(256), 0, stream, (const uint32_t*)lcs,
                               lens, pow_f32, kind, std::min(n, r1), (size_t)a.ld, a.D, tile0, off);
    } else {
        const int first = std::max(r0, 1);
        if (r1 <= first) return hipSuccess;
        if (elem_size == 2)
            hipLaunchKernelGGL(upgma_dist_kernel<uint16_t>, dim3(r1 - first), dim3(256), 0, stream, (const uint16_t*)lcs, lens,
                               pow_f32, kind, first, off, a.D);
        else
            hipLaunchKernelGGL(upgma_dist_kernel<uint32_t>, dim3(r1 - first), dim3(256), 0, stream, (const uint32_t*)lcs, lens,
                               pow_f32, kind, first, off, a.D);
    }
    return hipGetLastError();
}

// every row's first strict minimum over its full row (after all distances are in place)
hipError_t launch_upgma_init(const UpgmaArgs& a, hipStream_t stream)
{
    if (a.square) hipLaunchKernelGGL(upgma_init_kernel<true>, dim3(a.n), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(upgma_init_kernel<false>, dim3(a.n), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_upgma_steps(const UpgmaArgs& a, bool modified, hipStream_t stream)
{
    const int n = a.n;
    hipLaunchKernelGGL(upgma_block_min_kernel, dim3(a.n_blocks), dim3(256), 0, stream, a);
    for (int it = 0; it < n; ++it) {
#define UPGMA_STEP(M, S) hipLaunchKernelGGL((upgma_step_kernel<M, S>), dim3(a.n_blocks), dim3(256), 0, stream, a, it)
        if (modified) { if (a.square) UPGMA_STEP(true, true); else UPGMA_STEP(true, false); }
        else { if (a.square) UPGMA_STEP(false, true); else UPGMA_STEP(false, false); }
#undef UPGMA_STEP
    }
    return hipGetLastError();
}

} // namespace lcsgpu

// =============================================================================================
// Device-side neighbour joining, operation for operation the reference's
// NeighborJoining::computeTree (tree/NeighborJoining.cpp:33-118): float arithmetic with its
// exact association and summation ORDER (the sums of distances are accumulated sequentially in
// ascending cluster order, so they are here too), first strict minimum of
//   q(i,j) = (n_clusters - 2) * D[i,j] - sum[i] - sum[j]     over i < j in lexicographic order.
// The reference keeps its clusters in a vector it erases from; positions stay in ascending row
// order, so "position order" == "ascending row id among the active rows".
// Per merge: nj_rowmin (one workgroup per row j, scanning i < j: contiguous in the triangle),
// nj_select (1 workgroup: global first minimum, bookkeeping), nj_update (all k: new distances,
// sums of the other clusters), nj_sum (1 workgroup: the merged cluster's sum, in order).
// =============================================================================================
namespace lcsgpu {

__global__ __launch_bounds__(256) void nj_init_kernel(NjArgs a)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    float s = 0.0f;
    for (int j = 0; j < a.n; ++j)
        if (j != i) s = __fadd_rn(s, a.D[tri_index(i, j)]); // sequential, j ascending (NeighborJoining.cpp:47-53)
    a.sum[i] = s;
    a.node[i] = i;
    a.active[i] = 1;
}

__global__ __launch_bounds__(256) void nj_rowmin_kernel(NjArgs a, int n_clusters)
{
    __shared__ float s_q[256];
    __shared__ int s_i[256];
    const int j = blockIdx.x, tid = threadIdx.x;
    float bq = 3.40282347e38f; // numeric_limits<float>::max(): only q < max can win
    int bi = -1;
    if (a.active[j]) {
        const float sj = a.sum[j];
        const float f = (float)(n_clusters - 2);
        const float* row = a.D + (size_t)j * (j - 1) / 2;
        for (int i = tid; i < j; i += 256) {
            if (!a.active[i]) continue;
            const float q = __fsub_rn(__fsub_rn(__fmul_rn(f, row[i]), a.sum[i]), sj);
            if (q < bq) { bq = q; bi = i; } // ascending i within the thread
        }
    }
    s_q[tid] = bq;
    s_i[tid] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            const float q2 = s_q[tid + s];
            const int i2 = s_i[tid + s];
            if (i2 >= 0 && (s_i[tid] < 0 || q2 < s_q[tid] || (q2 == s_q[tid] && i2 < s_i[tid]))) { s_q[tid] = q2; s_i[tid] = i2; }
        }
        __syncthreads();
    }
    if (tid == 0) {
        a.part_q[j] = s_q[0];
        a.part_i[j] = s_i[0];
    }
}

__global__ __launch_bounds__(1024) void nj_select_kernel(NjArgs a, int iter)
{
    __shared__ float s_q[1024];
    __shared__ int s_i[1024], s_j[1024];
    const int tid = threadIdx.x;
    float bq = 3.40282347e38f;
    int bi = -1, bj = -1;
    for (int j = tid; j < a.n; j += 1024) {
        const int i = a.part_i[j];
        if (i < 0) continue;
        const float q = a.part_q[j];
        // first strict minimum in (i, j) lexicographic order
        if (bi < 0 || q < bq || (q == bq && (i < bi || (i == bi && j < bj)))) { bq = q; bi = i; bj = j; }
    }
    s_q[tid] = bq; s_i[tid] = bi; s_j[tid] = bj;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (tid < s) {
            const float q2 = s_q[tid + s];
            const int i2 = s_i[tid + s], j2 = s_j[tid + s];
            if (i2 >= 0 && (s_i[tid] < 0 || q2 < s_q[tid] ||
                            (q2 == s_q[tid] && (i2 < s_i[tid] || (i2 == s_i[tid] && j2 < s_j[tid]))))) {
                s_q[tid] = q2; s_i[tid] = i2; s_j[tid] = j2;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        int mi = s_i[0], mj = s_j[0];
        if (mi < 0) { // no q below FLT_MAX: the reference keeps min_i = min_j = 0 (degenerate input)
            a.sel[2] = 1;
            mi = 0; mj = 0;
        }
        a.sel[0] = mi;
        a.sel[1] = mj;
        a.left[iter] = a.node[mi];
        a.right[iter] = a.node[mj];
    }
}

__global__ __launch_bounds__(256) void nj_update_kernel(NjArgs a)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= a.n) return;
    const int mi = a.sel[0], mj = a.sel[1];
    float nd = 0.0f; // contribution to the merged cluster's sum (0 for clusters that do not take part)
    if (a.active[k] && k != mi && k != mj) {
        const size_t vi = tri_index(mi, k);
        const float Dij = a.D[tri_index(mi, mj)];
        const float Dik = a.D[vi], Djk = a.D[tri_index(mj, k)];
        float sk = __fsub_rn(a.sum[k], __fadd_rn(Dik, Djk));          // ck.sum -= Dik + Djk
        nd = __fmul_rn(__fsub_rn(__fadd_rn(Dik, Djk), Dij), 0.5f);    // (Dik + Djk - Dij) / 2
        sk = __fadd_rn(sk, nd);                                       // ck.sum += Dik
        a.sum[k] = sk;
        a.D[vi] = nd;
    }
    a.tmp[k] = nd;
}

__global__ __launch_bounds__(256) void nj_sum_kernel(NjArgs a, int iter)
{
    // ci.sum = sum of the new distances in ascending cluster order (NeighborJoining.cpp:88-108): a
    // sequential float sum.  Clusters that take no part contribute +0.0f, the identity (the sum starts
    // at +0.0f and cannot become -0.0f), so each wave first compacts the non-zero addends of its
    // quarter of a chunk, in order, and one lane then walks the four compacted runs.
    __shared__ float buf[4096];
    __shared__ float nz[4][1024];
    __shared__ int cnt[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    float s = 0.0f;
    for (int base = 0; base < a.n; base += 4096) {
        const int m = min(4096, a.n - base);
        for (int t = tid; t < 4096; t += 256) buf[t] = t < m ? a.tmp[base + t] : 0.0f;
        __syncthreads();
        int c = 0;
        for (int t0 = wave * 1024; t0 < wave * 1024 + 1024; t0 += 256) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = buf[t0 + 64 * u + lane];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint64_t mask = __ballot(v[u] != 0.0f);
                if (v[u] != 0.0f) nz[wave][c + __popcll(mask & lt_mask)] = v[u];
                c += __popcll(mask);
            }
        }
        if (lane == 0) cnt[wave] = c;
        __syncthreads();
        if (tid == 0) {
            for (int w = 0; w < 4; ++w) {
                const int k = cnt[w];
                int t = 0;
                for (; t + 8 <= k; t += 8) {
                    float g[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) g[u] = nz[w][t + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) s = __fadd_rn(s, g[u]);
                }
                for (; t < k; ++t) s = __fadd_rn(s, nz[w][t]);
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const int mi = a.sel[0], mj = a.sel[1];
        a.sum[mi] = s;
        a.node[mi] = a.n + iter;
        if (mj != mi) a.active[mj] = 0;
    }
}

__global__ void nj_final_kernel(NjArgs a, int iter)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int first = -1, second = -1;
    for (int k = 0; k < a.n; ++k)
        if (a.active[k]) {
            if (first < 0) first = k;
            else if (second < 0) second = k;
        }
    a.left[iter] = a.node[first];
    a.right[iter] = second >= 0 ? a.node[second] : a.node[first];
}

hipError_t launch_nj_init(const NjArgs& a, hipStream_t stream)
{
    hipLaunchKernelGGL(nj_init_kernel, dim3((a.n + 255) / 256), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_nj(const NjArgs& a, hipStream_t stream)
{
    const int n = a.n, blocks = (n + 255) / 256;
    hipLaunchKernelGGL(nj_init_kernel, dim3(blocks), dim3(256), 0, stream, a);
    int iter = 0;
    for (int n_clusters = n; n_clusters > 2; --n_clusters, ++iter) {
        hipLaunchKernelGGL(nj_rowmin_kernel, dim3(n), dim3(256), 0, stream, a, n_clusters);
        hipLaunchKernelGGL(nj_select_kernel, dim3(1), dim3(1024), 0, stream, a, iter);
        hipLaunchKernelGGL(nj_update_kernel, dim3(blocks), dim3(256), 0, stream, a);
        hipLaunchKernelGGL(nj_sum_kernel, dim3(1), dim3(256), 0, stream, a, iter);
    }
    hipLaunchKernelGGL(nj_final_kernel, dim3(1), dim3(64), 0, stream, a, iter);
    return hipGetLastError();
}

// float distance triangle for the reducers (shared by UPGMA and NJ)
hipError_t launch_float_distances(const void* lcs, int elem_size, const uint32_t* lens, const float* pow_f32, int kind,
                                  int n, float* D, hipStream_t stream)
{
    if (n < 2) return hipSuccess;
    if (elem_size == 2)
        hipLaunchKernelGGL(upgma_dist_kernel<uint16_t>, dim3(n - 1), dim3(256), 0, stream, (const uint16_t*)lcs, lens,
                           pow_f32, kind, 1, (size_t)0, D);
    else
        hipLaunchKernelGGL(upgma_dist_kernel<uint32_t>, dim3(n - 1), dim3(256), 0, stream, (const uint32_t*)lcs, lens,
                           pow_f32, kind, 1, (size_t)0, D);
    return hipGetLastError();
}

} // namespace lcsgpu

// =============================================================================================
// Seed assignment of the FastTree recursion (FastTree::makeEvaluation, tree/FastTree.cpp:309-324):
// for the seeds in order, d = Transform<float>(LCS(ref = seed, partner = column)); a column moves to a
// seed only on a strictly smaller distance.  One lane per column over the LCS rectangle the engine has
// just computed into HBM: 8 bytes per column leave the device instead of 2 bytes per pair.
// =============================================================================================
namespace lcsgpu {

template <typename T>
__global__ __launch_bounds__(256) void assign_seeds_kernel(const T* __restrict__ lcs, int64_t ld,
                                                           const int32_t* __restrict__ seed_ids, int32_t n_seeds,
                                                           const int32_t* __restrict__ col_ids, int32_t n_cols,
                                                           const uint32_t* __restrict__ lens,
                                                           const float* __restrict__ pow_f32, int kind, int first_k,
                                                           float* __restrict__ dist, int32_t* __restrict__ assign)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n_cols) return;
    const uint32_t len_j = lens[col_ids[j]];
    float best = dist[j];
    int32_t who = assign[j];
    for (int r = 0; r < n_seeds; ++r) {
        const uint32_t l = lcs[(int64_t)r * ld + j];
        const uint32_t indel = lens[seed_ids[r]] + len_j - 2u * l;
        float d;
        if (l == 0) d = 3.40282347e38f; // Transform<float>: (float) nextafter((double) FLT_MAX, 0) == FLT_MAX
        else if (kind == 1) d = __fdiv_rn(pow_f32[indel], (float)l);
        else d = __fdiv_rn((float)indel, (float)l);
        if (d < best) { best = d; who = first_k + r; }
    }
    dist[j] = best;
    assign[j] = who;
}

// The same for SEVERAL evaluations in one launch (lcsgpu_assign_seeds_batch), each from scratch: piece p = columns
// [col0, col0 + n_cols) of the launch's concatenated column list against the seeds [seed0, seed0 + n_seeds) of the
// concatenated seed list, its LCS rectangle (n_seeds x n_cols) at element out0.
template <typename T>
__global__ __launch_bounds__(256) void assign_seeds_batch_kernel(const T* __restrict__ lcs, const AssignPiece* __restrict__ pieces, int32_t n_pieces,
                                                                 const int32_t* __restrict__ seed_ids, const int32_t* __restrict__ col_ids,
                                                                 int64_t n_cols, const uint32_t* __restrict__ lens,
                                                                 const float* __restrict__ pow_f32, int kind,
                                                                 float* __restrict__ dist, int32_t* __restrict__ assign)
{
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= n_cols) return;
    int lo = 0, hi = n_pieces - 1; // the last piece that starts at or before c
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (pieces[mid].col0 <= c) lo = mid; else hi = mid - 1;
    }
    const AssignPiece p = pieces[lo];
    const uint32_t len_j = lens[col_ids[c]];
    const T* col = lcs + p.out0 + (c - p.col0);
    float best = __int_as_float(0x7f800000); // +inf: the first seed always takes the column (its distance is finite)
    int32_t who = 0;
    for (int r = 0; r < p.n_seeds; ++r) {
        const uint32_t l = col[(int64_t)r * p.n_cols];
        const uint32_t indel = lens[seed_ids[p.seed0 + r]] + len_j - 2u * l;
        float d;
        if (l == 0) d = 3.40282347e38f;
        else if (kind == 1) d = __fdiv_rn(pow_f32[indel], (float)l);
        else d = __fdiv_rn((float)indel, (float)l);
        if (d < best) { best = d; who = r; }
    }
    dist[c] = best;
    assign[c] = who;
}

hipError_t launch_assign_seeds_batch(const void* lcs, int elem_size, const AssignPiece* pieces, int32_t n_pieces, const int32_t* seed_ids,
                                     const int32_t* col_ids, int64_t n_cols, const uint32_t* lens, const float* pow_f32, int kind,
                                     float* dist, int32_t* assign, hipStream_t stream)
{
    if (n_cols <= 0 || n_pieces <= 0) return hipSuccess;
    const dim3 grid((unsigned)((n_cols + 255) / 256));
    if (elem_size == 2)
        hipLaunchKernelGGL(assign_seeds_batch_kernel<uint16_t>, grid, dim3(256), 0, stream, (const uint16_t*)lcs, pieces, n_pieces,
                           seed_ids, col_ids, n_cols, lens, pow_f32, kind, dist, assign);
    else
        hipLaunchKernelGGL(assign_seeds_batch_kernel<uint32_t>, grid, dim3(256), 0, stream, (const uint32_t*)lcs, pieces, n_pieces,
                           seed_ids, col_ids, n_cols, lens, pow_f32, kind, dist, assign);
    return hipGetLastError();
}

hipError_t launch_assign_seeds(const void* lcs, int elem_size, int64_t ld, const int32_t* seed_ids, int32_t n_seeds,
                               const int32_t* col_ids, int32_t n_cols, const uint32_t* lens, const float* pow_f32,
                               int kind, int first_k, float* dist, int32_t* assign, hipStream_t stream)
{
    if (n_cols <= 0 || n_seeds <= 0) return hipSuccess;
    const dim3 grid((unsigned)((n_cols + 255) / 256));
    if (elem_size == 2)
        hipLaunchKernelGGL(assign_seeds_kernel<uint16_t>, grid, dim3(256), 0, stream, (const uint16_t*)lcs, ld, seed_ids,
                           n_seeds, col_ids, n_cols, lens, pow_f32, kind, first_k, dist, assign);
    else
        hipLaunchKernelGGL(assign_seeds_kernel<uint32_t>, grid, dim3(256), 0, stream, (const uint32_t*)lcs, ld, seed_ids,
                           n_seeds, col_ids, n_cols, lens, pow_f32, kind, first_k, dist, assign);
    return hipGetLastError();
}

} // namespace lcsgpu
