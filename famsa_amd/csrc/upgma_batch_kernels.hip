// upgma_batch_kernels.hip -- UPGMA::computeTree (reference tree/UPGMA.cpp:198-288) with SEVERAL merges per launch.
//
// One launch per merge (tree_kernels.hip, upgma_step_kernel) costs a dependent launch per merge: 8.4 us each at
// 100 000 sequences for ~1.6 MB of traffic.  What a merge needs from the one before it is less than it looks:
//   * WHICH pair merges next depends on the previous merge only through ONE number.  The reference picks the first
//     minimum of min_dist[] over the active rows; min_dist of a row never changes except for the row a merge
//     creates (its other row dies), and nearest[] only changes by the rename Rmin -> Lmin.  So with the active rows
//     kept SORTED by (min_dist, index), the next picks are the next entries of that order -- skipping rows that died
//     on the way -- unless the key of a row created meanwhile is smaller.  That is checked afterwards (below); on
//     data it practically never is (a cluster's distances are averages: its minimum is larger than the small ends
//     of the order).
//   * The VALUES of a merge (new row = average of the two rows, column by column) depend on earlier merges of the
//     batch only inside the same column: thread j holds column j of every row the batch creates, so a merge
//     whose partner was created earlier in the batch (a cluster that keeps growing -- the rule in protein families,
//     where one hub sequence is everybody's nearest neighbour) reads that value from its own registers / LDS.
//     Only the columns of rows created IN the batch see each other (K x K "cross" entries): those are left to the
//     commit kernel, which has every column's result in front of it.
// A batch is two launches:
//   upgma_batch_rows_kernel    every workgroup walks the first entries of the sorted order (one wave, DPP / ballots:
//                              no LDS, no barrier) to the batch's <= K merges (L_t, R_t), then thread j computes
//                              column j of every merge from the COMMITTED matrix (all row loads issued at once)
//                              into side rows [K][n], with per-workgroup first minima of every new row.  Nothing
//                              of the committed state is written.
//   upgma_batch_commit_kernel  every workgroup: the new rows' minima (partials + cross entries), the validity
//                              prefix V -- merge t stands iff (key_t, L_t) < (new minimum, row) of every row
//                              created before it in the batch and still alive: exactly "the reference would have
//                              picked L_t" -- then for t < V: rows and mirror columns of the symmetric matrix from
//                              the side rows, the cross entries, nearest renames, min_dist / nearest / node_index
//                              of the merged rows, left / right; and the sorted order rewritten (entries of
//                              merged rows out, the new rows in at their keys, every entry one coalesced copy),
//                              with the first 2K entries and their nearest as the next batch's candidates.
// Merges t >= V are dropped (their side rows are never committed) and the next batch starts from the committed state,
// so the sequence of merges, every float operation and every tie rule are the reference's, whatever V is.
// Needs the full symmetric matrix (both rows of a merge contiguous); the packed triangle keeps one launch per merge.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dpp_min.h"
#include "lcs_kernels.h"

namespace lcsgpu {

static constexpr uint32_t UB_NONE = 0x7FFFFFFFu;
static constexpr float UB_BIG = 1e29f;            // UPGMA::BIG_DIST, reference tree/UPGMA.h:106
static constexpr uint32_t UB_BIG_BITS = 0x6FA18F08u; // its bit pattern: keys are compared as the bits of positive floats
static constexpr int UB_INF = 1 << 20;

template <bool MODIFIED>
__device__ __forceinline__ float ub_average(float x, float y)
{
    if (MODIFIED) // 0.05f * (x + y) + 0.9f * min(x, y), no contraction (reference UPGMA.cpp:32-34)
        return __fadd_rn(__fmul_rn(0.05f, __fadd_rn(x, y)), __fmul_rn(0.9f, fminf(x, y)));
    return __fmul_rn(__fadd_rn(x, y), 0.5f); // (x + y) * 0.5f
}

__device__ __forceinline__ void ub_take(float d, uint32_t j, float& bd, uint32_t& bj)
{
    if (d < UB_BIG && (d < bd || (d == bd && j < bj))) { bd = d; bj = j; }
}
__device__ __forceinline__ bool ub_less(uint32_t k1, uint32_t r1, uint32_t k2, uint32_t r2) // (key bits, row) order
{
    return k1 < k2 || (k1 == k2 && r1 < r2);
}
__device__ __forceinline__ uint32_t lane_u32(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }
__device__ __forceinline__ int lane_i32(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

// ---- the sorted order, once: rank of every row's (min_dist, row) by counting (n^2 / 2 compares of keys held in LDS:
// a millisecond at 100 000 rows, next to the 30 ms of the distance prologue) ------------------------------------------
__global__ __launch_bounds__(256) void upgma_batch_rank_kernel(UpgmaBatchArgs a)
{
    __shared__ uint32_t s_k[1024];
    const int n = a.n;
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    const uint32_t my = j < (uint32_t)n ? __float_as_uint(a.min_dist[j]) : 0xFFFFFFFFu;
    uint32_t rank = 0;
    for (int base = 0; base < n; base += 1024) {
        __syncthreads();
        for (int x = threadIdx.x; x < 1024; x += 256) s_k[x] = base + x < n ? __float_as_uint(a.min_dist[base + x]) : 0xFFFFFFFFu;
        __syncthreads();
        const int lim = min(1024, n - base);
        for (int x = 0; x < lim; ++x) {
            const uint32_t k = s_k[x];
            rank += (k < my || (k == my && (uint32_t)(base + x) < j)) ? 1u : 0u;
        }
    }
    if (j < (uint32_t)n) {
        a.sorted0[rank] = make_uint2(my, j);
        a.pos[j] = rank;
        if (rank < (uint32_t)UPGMA_BATCH_CAND) a.cand[rank] = make_uint4(my, j, a.nearest[j], 0u);
    }
    if (j == 0) {
        a.state[0] = 0u;          // merges committed
        a.state[1] = (uint32_t)n; // entries of the sorted order = active rows
        a.state[2] = 0u;          // error: no finite nearest neighbour
        a.state[3] = 0u;          // batches that were cut short by the validity check (statistics)
        a.hdr[0] = 0u;
    }
}

// hdr: [0] = m (merges of the pending batch), [1] = error seen by the walk, then per merge t, at 8 + 8 t:
//   L, R, key bits, src (merge of this batch that created R, or -1), position of L in the order, position of R (or NONE)
constexpr int UB_HDR0 = 8, UB_HDR_STRIDE = 8;

// ---- launch 1 of a batch ------------------------------------------------------------------------------------------------
template <int K, bool MODIFIED>
__global__ __launch_bounds__(256) void upgma_batch_rows_kernel(UpgmaBatchArgs a, int parity)
{
    static_assert(2 * K <= 64, "the candidates of a batch sit in the lanes of one wave");
    __shared__ float s_new[K][256]; // column tid of every row the batch creates (read back by the same thread only)
    __shared__ float s_pd[K][4];
    __shared__ uint32_t s_pj[K][4];
    const int tid = threadIdx.x, b = blockIdx.x, n = a.n, nb = a.n_blocks;
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t j = (uint32_t)b * 256 + tid;
    const bool in = j < (uint32_t)n;
    // ---- level 1: everything whose address is known before the launch ----
    const uint32_t* st = a.state + 8 * parity;
    const uint32_t done = st[0], ns = st[1], err = st[2];
    const uint32_t my_node = in ? a.node_index[j] : UB_NONE;
    uint4 c = make_uint4(0xFFFFFFFFu, UB_NONE, UB_NONE, 0u);
    if (lane < 2 * K) c = a.cand[lane];
    const int n_cand = (int)min(ns, (uint32_t)(2 * K));
    if (done >= (uint32_t)(n - 1) || err) {
        if (b == 0 && tid == 0) a.hdr[0] = 0u;
        return;
    }
    // ---- the walk: lane t of every wave ends up holding merge t ----
    uint32_t mL = UB_NONE, mR = UB_NONE, mKey = 0u;
    int mSrc = -1, mPos = 0;
    int m = 0;
    bool bad = false;
    const int budget = (int)min((uint32_t)K, (uint32_t)(n - 1) - done);
    for (int i = 0; i < n_cand && m < budget; ++i) {
        const uint32_t L = lane_u32(c.y, i), key = lane_u32(c.x, i);
        uint32_t R = lane_u32(c.z, i);
        if (__ballot(lane < m && mR == L)) continue; // this row died earlier in the batch
        if (key >= UB_BIG_BITS) { bad = true; break; } // no row left with a finite nearest neighbour (reference: undefined)
        for (;;) { // renames of the batch: a row that died became the row it merged into
            const unsigned long long hit = __ballot(lane < m && mR == R);
            if (!hit) break;
            R = lane_u32(mL, __builtin_ctzll(hit));
        }
        if (R == UB_NONE || R >= (uint32_t)n) { bad = true; break; }
        const unsigned long long made = __ballot(lane < m && mL == R);
        const int src = made ? (int)__builtin_ctzll(made) : -1;
        if (lane == m) { mL = L; mR = R; mKey = key; mSrc = src; mPos = i; }
        ++m;
    }
    if (b == 0 && wave == 0) { // the batch's record for the commit kernel
        if (lane == 0) { a.hdr[0] = (uint32_t)m; a.hdr[1] = bad ? 1u : 0u; }
        if (lane < m) {
            uint32_t* h = a.hdr + UB_HDR0 + UB_HDR_STRIDE * lane;
            h[0] = mL; h[1] = mR; h[2] = mKey; h[3] = (uint32_t)mSrc; h[4] = (uint32_t)mPos;
            h[5] = mSrc < 0 ? a.pos[mR] : UB_NONE;
        }
    }
    if (m == 0) return;
    // ---- my column's part in the batch ----
    int createAt = UB_INF, dieAt = UB_INF;
#pragma unroll
    for (int t = 0; t < K; ++t) {
        if (t < m) {
            if (lane_u32(mL, t) == j) createAt = t;
            if (lane_u32(mR, t) == j && dieAt == UB_INF) dieAt = t;
        }
    }
    const bool alive0 = in && my_node != UB_NONE;
    float dl[K], dr[K];
#pragma unroll
    for (int t = 0; t < K; ++t) { // every row load of the batch, issued together
        dl[t] = 0.0f;
        dr[t] = 0.0f;
        if (t < m) {
            const bool normal = alive0 && createAt > t && dieAt > t;
            const uint32_t Lt = lane_u32(mL, t), Rt = lane_u32(mR, t);
            const int src = lane_i32(mSrc, t);
            if (normal) {
                dl[t] = a.D[(size_t)Lt * (size_t)n + j];
                if (src < 0) dr[t] = a.D[(size_t)Rt * (size_t)n + j];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < K; ++t) {
        if (t < m) {
            const bool normal = alive0 && createAt > t && dieAt > t;
            const int src = lane_i32(mSrc, t);
            float nd = UB_BIG;
            uint32_t nj = UB_NONE;
            if (normal) {
                const float dR = src < 0 ? dr[t] : s_new[src][tid];
                const float v = ub_average<MODIFIED>(dl[t], dR);
                s_new[t][tid] = v;
                a.side[(size_t)t * (size_t)n + j] = v;
                ub_take(v, j, nd, nj);
            }
            wave_first_min(nd, nj);
            if (lane == 0) { s_pd[t][wave] = nd; s_pj[t][wave] = nj; }
        }
    }
    __syncthreads();
    if (tid < m) {
        float d = UB_BIG;
        uint32_t dj = UB_NONE;
#pragma unroll
        for (int w = 0; w < 4; ++w) ub_take(s_pd[tid][w], s_pj[tid][w], d, dj);
        a.part_d[(size_t)tid * nb + b] = d;
        a.part_j[(size_t)tid * nb + b] = dj;
    }
}

// ---- launch 2 of a batch ------------------------------------------------------------------------------------------------
template <int K, bool MODIFIED>
__global__ __launch_bounds__(256) void upgma_batch_commit_kernel(UpgmaBatchArgs a, int parity)
{
    __shared__ float s_side[K][2 * K]; // side row u at the columns of the batch's rows: [u][k] = L_k, [u][K + k] = R_k
    __shared__ float s_tab[K][K];      // cross entries: [t][u] = D[L_t][L_u] right after merge t (u < t)
    __shared__ float s_rd[256];
    __shared__ uint32_t s_rj[256];
    __shared__ float s_pm_d[K];
    __shared__ uint32_t s_pm_j[K];
    __shared__ uint32_t s_L[K], s_R[K], s_near[K], s_minbits[K], s_posL[K], s_posR[K];
    __shared__ int s_src[K], s_die[K];
    __shared__ int s_V;
    const int tid = threadIdx.x, b = blockIdx.x, n = a.n, nb = a.n_blocks;
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t gid = (uint32_t)b * 256 + tid; // row j AND position p of the sorted order
    const bool in = gid < (uint32_t)n;
    const uint32_t* st = a.state + 8 * parity;
    uint32_t* st_next = a.state + 8 * (parity ^ 1);
    // ---- level 1 ----
    const uint32_t done = st[0], ns = st[1], err = st[2], cuts = st[3];
    const int m = (int)a.hdr[0];
    const uint32_t walk_err = a.hdr[1];
    uint32_t mL = UB_NONE, mR = UB_NONE, mKey = 0u, mPosL = UB_NONE, mPosR = UB_NONE;
    int mSrc = -1;
    if (lane < K) { // (unconditional: entries beyond m are stale words of an earlier batch, never used)
        const uint32_t* h = a.hdr + UB_HDR0 + UB_HDR_STRIDE * lane;
        mL = h[0]; mR = h[1]; mKey = h[2]; mSrc = (int)h[3]; mPosL = h[4]; mPosR = h[5];
    }
    const uint32_t my_node = in ? a.node_index[gid] : UB_NONE;
    const uint32_t my_near = in ? a.nearest[gid] : UB_NONE;
    const uint2* cur = parity ? a.sorted1 : a.sorted0;
    uint2* nxt = parity ? a.sorted0 : a.sorted1;
    uint2 e = make_uint2(0xFFFFFFFFu, UB_NONE), ep = make_uint2(0u, 0u);
    if (gid < ns) e = cur[gid];
    if (gid > 0 && gid <= ns) ep = cur[gid - 1];
    float sv[K];
#pragma unroll
    for (int t = 0; t < K; ++t) sv[t] = in ? a.side[(size_t)t * (size_t)n + gid] : 0.0f;
    // the partial minima of the new rows: 256 / K threads per merge
    constexpr int TPM = 256 / K;
    {
        const int t = tid / TPM, sub = tid % TPM;
        float d = UB_BIG;
        uint32_t dj = UB_NONE;
        for (int x = sub; x < nb; x += TPM) ub_take(a.part_d[(size_t)t * nb + x], a.part_j[(size_t)t * nb + x], d, dj);
        s_rd[tid] = d;
        s_rj[tid] = dj;
    }
    if (m == 0 || err) { // nothing pending (finished, or an error): the state moves on unchanged
        if (b == 0 && tid == 0) {
            st_next[0] = done; st_next[1] = ns; st_next[2] = err | walk_err; st_next[3] = cuts;
        }
        return;
    }
    if (tid < K) {
        s_L[tid] = mL; s_R[tid] = mR; s_src[tid] = mSrc; s_posL[tid] = mPosL; s_posR[tid] = mPosR;
    }
    __syncthreads();
    // ---- level 2: side rows at the batch's own columns; the nodes of the merged rows ----
    for (int idx = tid; idx < K * 2 * K; idx += 256) {
        const int u = idx / (2 * K), k = idx % (2 * K);
        const int t = k < K ? k : k - K;
        float v = 0.0f;
        if (u < m && t < m) {
            const uint32_t x = k < K ? s_L[t] : s_R[t];
            v = a.side[(size_t)u * (size_t)n + x];
        }
        s_side[u][k] = v;
    }
    uint32_t nodeL = UB_NONE, nodeR = UB_NONE;
    if (b == 0 && wave == 0 && lane < m) {
        nodeL = a.node_index[mL];
        nodeR = mSrc < 0 ? a.node_index[mR] : UB_NONE;
    }
    if (tid < K) {
        float d = UB_BIG;
        uint32_t dj = UB_NONE;
        for (int s = 0; s < TPM; ++s) ub_take(s_rd[tid * TPM + s], s_rj[tid * TPM + s], d, dj);
        s_pm_d[tid] = d;
        s_pm_j[tid] = dj;
    }
    __syncthreads();
    // ---- the cross entries, the new rows' minima, the validity prefix: wave 0 ----
    if (wave == 0) {
        // die[u]: the merge at which the row created by merge u dies again (it is some later merge's R), or INF
        int die = UB_INF;
        for (int w = 0; w < m; ++w) {
            const uint32_t Rw = lane_u32(mR, w);
            if (lane < w && lane < m && mL == Rw && die == UB_INF) die = w;
        }
        float newmin = UB_BIG;
        uint32_t newnear = UB_NONE;
        for (int t = 0; t < m; ++t) {
            const int src_t = lane_i32(mSrc, t);
            float cd = UB_BIG;
            uint32_t cj = UB_NONE;
            if (lane < t && die > t) { // column L_lane is alive at merge t: a cross entry
                const int u = lane;
                const float a1 = s_side[u][t]; // D[L_t][L_u] before merge t: row L_u (made by merge u) at column L_t
                float a2;
                if (src_t < 0) a2 = s_side[u][K + t];
                else a2 = src_t > u ? s_tab[src_t][u] : s_tab[u][src_t];
                const float v = ub_average<MODIFIED>(a1, a2);
                s_tab[t][u] = v;
                ub_take(v, mL, cd, cj);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            wave_first_min(cd, cj);
            float d = s_pm_d[t];
            uint32_t dj = s_pm_j[t];
            ub_take(cd, cj, d, dj); // (first minimum: smaller value, then smaller column)
            if (d >= UB_BIG) { d = UB_BIG; dj = UB_NONE; }
            if (lane == t) { newmin = d; newnear = dj; }
        }
        // merge t stands iff the reference would have picked L_t: its key beats every row created before it in the
        // batch that is still alive at that point
        bool ok = true;
        for (int s = 0; s < m; ++s) {
            const float ms = __uint_as_float(lane_u32(__float_as_uint(newmin), s));
            const uint32_t Ls = lane_u32(mL, s);
            const int ds = lane_i32(die, s);
            if (s < lane && ds >= lane) { // (alive when merge `lane` is picked -- also as that merge's own partner)
                const float mk = __uint_as_float(mKey);
                if (!(mk < ms || (mk == ms && mL < Ls))) ok = false;
            }
        }
        const unsigned long long fail = __ballot(lane < m && !ok);
        const int V = fail ? (int)__builtin_ctzll(fail) : m;
        // the new rows' nearest under the later renames of the batch
        for (int w = 0; w < V; ++w) {
            const uint32_t Rw = lane_u32(mR, w), Lw = lane_u32(mL, w);
            if (w > lane && newnear == Rw) newnear = Lw;
        }
        if (lane < K) {
            s_near[lane] = newnear;
            s_minbits[lane] = __float_as_uint(newmin);
            s_die[lane] = die;
        }
        if (lane == 0) s_V = V;
        if (b == 0 && lane < V) { // the merged rows' bookkeeping (UPGMA.cpp:268-287)
            a.left[done + lane] = (int32_t)nodeL;
            a.right[done + lane] = (int32_t)(mSrc < 0 ? nodeR : (uint32_t)n + done + (uint32_t)mSrc);
            const bool dies = die < V;
            a.node_index[mL] = dies ? UB_NONE : (uint32_t)n + done + (uint32_t)lane;
            if (mSrc < 0) a.node_index[mR] = UB_NONE;
            if (!dies) {
                a.min_dist[mL] = newmin;
                a.nearest[mL] = newnear;
            }
        }
        if (b == 0 && lane == 0) {
            st_next[0] = done + (uint32_t)V;
            st_next[1] = ns - (uint32_t)V;
            st_next[2] = err | walk_err;
            st_next[3] = cuts + (V < m ? 1u : 0u);
        }
    }
    __syncthreads();
    const int V = s_V;
    // ---- commit my row / column ----
    int createAt = UB_INF, dieAt = UB_INF;
    for (int t = 0; t < V; ++t) {
        if (s_L[t] == gid) createAt = t;
        if (s_R[t] == gid && dieAt == UB_INF) dieAt = t;
    }
    const bool alive0 = in && my_node != UB_NONE;
    if (alive0 && createAt == UB_INF && dieAt == UB_INF) { // a row the batch only passes through
#pragma unroll
        for (int t = 0; t < K; ++t) {
            if (t < V) {
                const uint32_t Lt = s_L[t];
                a.D[(size_t)Lt * (size_t)n + gid] = sv[t];
                a.D[(size_t)gid * (size_t)n + Lt] = sv[t]; // the mirror
            }
        }
        uint32_t near = my_near;
        for (int t = 0; t < V; ++t)
            if (near == s_R[t]) near = s_L[t];
        if (near != my_near) a.nearest[gid] = near;
    }
    // (rows that take part in a standing merge: their entries towards earlier merges' rows are superseded by the cross
    //  entries or belong to a dead row; what they keep is written below / by workgroup 0 above)
    if (b == 0) {
        for (int idx = tid; idx < K * K; idx += 256) {
            const int t = idx / K, u = idx % K;
            if (u < t && t < V && s_die[u] > t) {
                const float v = s_tab[t][u];
                a.D[(size_t)s_L[t] * (size_t)n + s_L[u]] = v;
                a.D[(size_t)s_L[u] * (size_t)n + s_L[t]] = v;
            }
        }
    }
    // ---- the sorted order of the next batch: position gid of the current one (gid == ns: the place behind the end) ----
    if (gid <= ns) {
        bool removed = gid == ns;
        uint32_t before = 0; // entries in front of me that leave
        for (int t = 0; t < V; ++t) {
            const uint32_t pl = s_posL[t], pr = s_posR[t];
            removed = removed || pl == gid || pr == gid;
            before += (pl < gid ? 1u : 0u) + (pr != UB_NONE && pr < gid ? 1u : 0u);
        }
        uint32_t ins_before = 0;
        for (int s = 0; s < V; ++s) { // rows the batch created and that are still alive: they enter at their new keys
            if (s_die[s] < V) continue;
            const uint32_t kb = s_minbits[s], row = s_L[s];
            if (ub_less(kb, row, e.x, e.y)) { // in front of me (my slot e is +inf at gid == ns)
                ++ins_before;
                if (gid == 0 || !ub_less(kb, row, ep.x, ep.y)) { // ... and not in front of my predecessor: I place it
                    uint32_t at = gid - before;
                    for (int s2 = 0; s2 < V; ++s2)
                        if (s2 != s && s_die[s2] >= V && ub_less(s_minbits[s2], s_L[s2], kb, row)) ++at;
                    nxt[at] = make_uint2(kb, row);
                    a.pos[row] = at;
                    if (at < (uint32_t)UPGMA_BATCH_CAND) a.cand[at] = make_uint4(kb, row, s_near[s], 0u);
                }
            }
        }
        if (!removed) {
            const uint32_t at = gid - before + ins_before;
            nxt[at] = e;
            a.pos[e.y] = at;
            if (at < (uint32_t)UPGMA_BATCH_CAND) {
                uint32_t near = a.nearest[e.y]; // (a row the batch only passed through: its stored nearest is the old one)
                for (int t = 0; t < V; ++t)
                    if (near == s_R[t]) near = s_L[t];
                a.cand[at] = make_uint4(e.x, e.y, near, 0u);
            }
        }
    }
}

// ---- host side ------------------------------------------------------------------------------------------------------------
hipError_t launch_upgma_batch_init(const UpgmaBatchArgs& a, hipStream_t stream)
{
    hipLaunchKernelGGL(upgma_batch_rank_kernel, dim3(a.n_blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

// `count` batches (two launches each), the first of them batch number `first` (its parity selects the state buffers)
hipError_t launch_upgma_batches(const UpgmaBatchArgs& a, bool modified, int k, int first, int count, hipStream_t stream)
{
    const dim3 grid_rows(a.n_blocks), grid_commit((unsigned)(a.n / 256 + 1)), block(256);
    for (int i = first; i < first + count; ++i) {
        const int parity = i & 1;
#define UB_LAUNCH(KK)                                                                                                     \
    do {                                                                                                                  \
        if (modified) {                                                                                                   \
            hipLaunchKernelGGL((upgma_batch_rows_kernel<KK, true>), grid_rows, block, 0, stream, a, parity);              \
            hipLaunchKernelGGL((upgma_batch_commit_kernel<KK, true>), grid_commit, block, 0, stream, a, parity);          \
        } else {                                                                                                          \
            hipLaunchKernelGGL((upgma_batch_rows_kernel<KK, false>), grid_rows, block, 0, stream, a, parity);             \
            hipLaunchKernelGGL((upgma_batch_commit_kernel<KK, false>), grid_commit, block, 0, stream, a, parity);         \
        }                                                                                                                 \
    } while (0)
        if (k >= 32) UB_LAUNCH(32);
        else if (k >= 16) UB_LAUNCH(16);
        else UB_LAUNCH(8);
#undef UB_LAUNCH
    }
    return hipGetLastError();
}

} // namespace lcsgpu
