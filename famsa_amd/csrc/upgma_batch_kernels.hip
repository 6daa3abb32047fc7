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
// A batch is three launches:
//   upgma_batch_rows_kernel    every workgroup walks the first entries of the sorted order (one wave, readlanes / ballots:
//                              no LDS, no barrier) to the batch's <= K merges (L_t, R_t), then thread p computes slot
//                              p's entry of every merge from the COMMITTED matrix (all row loads issued at once)
//                              into side rows [K][slots], with per-workgroup first minima of every new row.  Nothing
//                              of the committed state is written.
//   upgma_batch_resolve_kernel ONE workgroup: the new rows' minima (partials + the cross entries between the batch's
//                              own clusters), the validity prefix V -- merge t stands iff (key_t, L_t) < (new minimum,
//                              row) of every row created before it in the batch and still alive: exactly "the
//                              reference would have picked L_t" -- and for t < V the merged rows' bookkeeping
//                              (left / right, node_index, min_dist, nearest, slots).  What it finds goes into a
//                              record every workgroup of the commit reads.
//   upgma_batch_commit_kernel  for t < V: the clusters' rows along the slots and every surviving row's run of V new
//                              entries from the side rows, the cross entries, nearest renames; and the sorted order
//                              rewritten (entries of merged rows out, the new rows in at their keys, every entry one
//                              coalesced copy), with its first 64 entries and their nearest as the next batch's candidates.
// Merges t >= V are dropped (their side rows are never committed) and the next batch starts from the committed state,
// so the sequence of merges, every float operation and every tie rule are the reference's, whatever V is.
//
// LAYOUT: rows and SLOTS.  The reference lets a new cluster take over its left child's row AND column of the matrix;
// writing that column is n scattered 4-byte stores per merge, each to a different page of a 40 GB matrix -- 3 of the 4 us
// a merge cost in the first batched form.  Here a cluster keeps its left child's ROW (row indices decide ties, so they
// stay the reference's) but gets a NEW COLUMN: the next free slot.  The matrix is D[row][slot], n rows x ld slots;
// slot_of[row] / row_of[slot] translate (a slot whose row died or moved is dead: row_of = NONE).  The
// columns of the V clusters a batch commits are then V CONSECUTIVE slots: every surviving row gets one contiguous run
// of V floats (128 B at V = 32) instead of V scattered words, and a cluster's own row is written along the slots
// (coalesced).  Thread p of the launches stands for slot p; first minima stay (value, ROW) so that ties resolve by row
// index exactly as the reference's ascending scans do.
// COMPACTION (round 5).  Every merge kills two slots and makes one, so at most n slots are ever alive -- rounds 4's
// n x 2n matrix (80 GB at 100 000 sequences) held mostly dead columns by the end.  Now ld = n + a spare of ~n/10 slots
// (44 GB at 100 000; the batched form reaches ~255 000 sequences on 288 GB): when the spare is used up the live slots
// move to the front in order (upgma_compact_map_kernel: remap[], row_of / slot_of rewritten, the next free slot = the
// number of live clusters) and every live row is packed in place (upgma_compact_rows_kernel: chunk by chunk, read -
// barrier - write; a value only ever moves to a smaller index).  The spare doubles with every compaction (the live
// clusters halve the other way), so a tree needs about log2(n / spare) of them -- 3 at n/10 -- each one sweep over what
// is left of the matrix (~15 ms the first at 100 000), and the batches after it cover fewer slots.  A cluster's node id
// (n + k for the k-th merge) is no longer its slot.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

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
        a.slot_of[j] = j; // leaf j sits in slot j
        if (rank < (uint32_t)UPGMA_BATCH_CAND) a.cand[rank] = make_uint4(my, j, a.nearest[j], 0u);
    }
    for (uint32_t p = j; p < (uint32_t)a.ld; p += gridDim.x * 256) a.row_of[p] = p < (uint32_t)n ? p : UB_NONE;
    if (j == 0) {
        a.state[0] = 0u;          // merges committed
        a.state[1] = (uint32_t)n; // entries of the sorted order = active rows
        a.state[2] = 0u;          // error: no finite nearest neighbour
        a.state[3] = 0u;          // batches that were cut short by the validity check (statistics)
        a.state[4] = (uint32_t)n; // the next free slot
        a.hdr[0] = 0u;
    }
}

// hdr: [0] = m (merges of the pending batch), [1] = error seen by the walk, then per merge t, at 8 + 8 t:
//   L, R, key bits, src (merge of this batch that created R, or -1), position of L in the order, position of R (or NONE),
//   slot of L, slot of R (or NONE when R was created in this batch)
constexpr int UB_HDR0 = 8, UB_HDR_STRIDE = 8;

// ---- launch 1 of a batch ------------------------------------------------------------------------------------------------
template <int K, bool MODIFIED>
__global__ __launch_bounds__(256) void upgma_batch_rows_kernel(UpgmaBatchArgs a, int parity)
{
    static_assert(2 * K <= 64, "the candidates of a batch sit in the lanes of one wave");
    constexpr int STRIDE = 256 + 8; // (the minima below read 8 rows of this at once: a stride of 8 banks keeps them apart)
    __shared__ float s_new[K][STRIDE]; // slot tid's entry of every row the batch creates (BIG where the slot takes no part)
    __shared__ uint32_t s_x[256];
    const int tid = threadIdx.x, b = blockIdx.x, n = a.n, nb = a.n_blocks;
    const size_t ld = (size_t)a.ld;
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t p = (uint32_t)b * 256 + tid; // my slot
    // ---- level 1: everything whose address is known before the launch ----
    const uint32_t* st = a.state + 8 * parity;
    const uint32_t done = st[0], ns = st[1], err = st[2];
    const uint32_t x = p < (uint32_t)a.ld ? a.row_of[p] : UB_NONE; // the row my slot stands for (NONE: dead / not yet used)
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
        // no row left in the order with a finite nearest neighbour: an error (the reference's behaviour is undefined) only
        // if the batch is empty -- the clusters of the m merges before it may get finite keys, and the reference, which
        // would pick those next, goes on: commit the prefix, let the next batch see the new rows
        if (key >= UB_BIG_BITS) { bad = m == 0; break; }
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
            h[6] = a.slot_of[mL];
            h[7] = mSrc < 0 ? a.slot_of[mR] : UB_NONE;
        }
    }
    if (m == 0) return;
    // ---- my slot's part in the batch: it is a column of merge t until its row is merged itself ----
    int dieAt = UB_INF;
#pragma unroll
    for (int t = 0; t < K; ++t)
        if (t < m && dieAt == UB_INF && (lane_u32(mL, t) == x || lane_u32(mR, t) == x)) dieAt = t;
    if (x == UB_NONE) dieAt = -1;
    float dl[K], dr[K];
#pragma unroll
    for (int t = 0; t < K; ++t) { // every row load of the batch, issued together
        dl[t] = 0.0f;
        dr[t] = 0.0f;
        if (t < m && dieAt > t) {
            dl[t] = a.D[(size_t)lane_u32(mL, t) * ld + p];
            if (lane_i32(mSrc, t) < 0) dr[t] = a.D[(size_t)lane_u32(mR, t) * ld + p];
        }
    }
    s_x[tid] = x;
#pragma unroll
    for (int t = 0; t < K; ++t) {
        if (t < m) {
            const int src = lane_i32(mSrc, t);
            float v = UB_BIG;
            if (dieAt > t) {
                const float dR = src < 0 ? dr[t] : s_new[src][tid];
                v = ub_average<MODIFIED>(dl[t], dR);
                a.side[(size_t)t * ld + p] = v;
            }
            s_new[t][tid] = v;
        }
    }
    __syncthreads();
    // first minimum of every new row over this workgroup's slots, ties by ROW index (the reference scans the rows in
    // ascending order): 256 / K threads per merge read its 256 entries back, then a DPP minimum inside the group
    constexpr int TPM = 256 / K; // 8, 16 or 32 consecutive lanes
    {
        const int t = tid / TPM, sub = tid % TPM;
        float nd = UB_BIG;
        uint32_t nj = UB_NONE;
        if (t < m) {
#pragma unroll 8
            for (int q = 0; q < K; ++q) ub_take(s_new[t][q * TPM + sub], s_x[q * TPM + sub], nd, nj);
        }
        dpp_min_step<DPP_QUAD_1032, 0xF>(nd, nj);
        dpp_min_step<DPP_QUAD_2301, 0xF>(nd, nj);
        dpp_min_step<DPP_ROW_HALF_MIRROR, 0xF>(nd, nj); // 8 lanes
        if (TPM >= 16) dpp_min_step<DPP_ROW_MIRROR, 0xF>(nd, nj);
        if (TPM >= 32) dpp_min_step<DPP_ROW_BCAST15, 0xA>(nd, nj); // result in the last lane of the group
        const bool holder = TPM >= 32 ? sub == TPM - 1 : sub == 0;
        if (t < m && holder) {
            a.part_d[(size_t)t * nb + b] = nd;
            a.part_j[(size_t)t * nb + b] = nj;
        }
    }
}

// ---- launch 2 of a batch: ONE workgroup resolves it ---------------------------------------------------------------------
// What every workgroup of the commit needs but only one has to work out: the new rows' minima (the rows kernel's
// per-workgroup partials + the K x K cross entries between the clusters the batch creates), the validity prefix V, the
// merged rows' bookkeeping.  Result record `rec` (words): [0] = V, [8 + 4t ..] = (new min_dist bits, new nearest, die)
// per merge, [8 + 4K ..] = the cross entries tab[t][u] as floats.
constexpr int UB_REC0 = 8;
template <int K, bool MODIFIED>
__global__ __launch_bounds__(1024) void upgma_batch_resolve_kernel(UpgmaBatchArgs a, int parity, int grid_rows)
{
    __shared__ float s_side[K][2 * K]; // side row u at the slots of the batch's rows: [u][k] = L_k, [u][K + k] = R_k
    __shared__ float s_tab[K][K + 1];  // cross entries: [t][u] = D[L_t][L_u] right after merge t (u < t)
    __shared__ float s_pm_d[K], s_min[K];
    __shared__ uint32_t s_pm_j[K], s_near[K];
    __shared__ uint32_t s_L[K], s_R[K], s_slotL[K], s_slotR[K];
    __shared__ int s_src[K], s_die[K];
    const int tid = threadIdx.x, n = a.n, nb = a.n_blocks;
    const size_t ld = (size_t)a.ld;
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t* st = a.state + 8 * parity;
    uint32_t* st_next = a.state + 8 * (parity ^ 1);
    uint32_t* rec = a.rec;
    // ---- level 1 ----
    const uint32_t done = st[0], ns = st[1], err = st[2], cuts = st[3], sb = st[4]; // sb: the next free slot
    const int m = (int)a.hdr[0];
    const uint32_t walk_err = a.hdr[1];
    uint32_t mL = UB_NONE, mR = UB_NONE, mKey = 0u, mSlotL = UB_NONE, mSlotR = UB_NONE;
    int mSrc = -1;
    if (lane < K) { // (unconditional: entries beyond m are stale words of an earlier batch, never used)
        const uint32_t* h = a.hdr + UB_HDR0 + UB_HDR_STRIDE * lane;
        mL = h[0]; mR = h[1]; mKey = h[2]; mSrc = (int)h[3]; mSlotL = h[6]; mSlotR = h[7];
    }
    { // the partial minima of the new rows: 32 lanes per merge, the loads of a lane in flight together
        const int t = tid >> 5, sub = tid & 31;
        float d = UB_BIG;
        uint32_t dj = UB_NONE;
        if (t < K) {
            const float* pd = a.part_d + (size_t)t * nb;
            const uint32_t* pj = a.part_j + (size_t)t * nb;
            for (int x0 = sub; x0 < grid_rows; x0 += 32 * 8) {
                float vd[8];
                uint32_t vj[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int xx = x0 + 32 * q;
                    vd[q] = xx < grid_rows ? pd[xx] : UB_BIG;
                    vj[q] = xx < grid_rows ? pj[xx] : UB_NONE;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) ub_take(vd[q], vj[q], d, dj); // (ascending x: any order gives the same first minimum)
            }
        }
        dpp_min_step<DPP_QUAD_1032, 0xF>(d, dj);
        dpp_min_step<DPP_QUAD_2301, 0xF>(d, dj);
        dpp_min_step<DPP_ROW_HALF_MIRROR, 0xF>(d, dj);
        dpp_min_step<DPP_ROW_MIRROR, 0xF>(d, dj);
        dpp_min_step<DPP_ROW_BCAST15, 0xA>(d, dj);
        if (t < K && sub == 31) { s_pm_d[t] = d; s_pm_j[t] = dj; }
    }
    if (m == 0 || err) { // nothing pending (finished, or an error): the state moves on unchanged
        if (tid == 0) {
            st_next[0] = done; st_next[1] = ns; st_next[2] = err | walk_err; st_next[3] = cuts; st_next[4] = sb;
            rec[0] = 0u;
        }
        return;
    }
    if (tid < K) {
        s_L[tid] = mL; s_R[tid] = mR; s_src[tid] = mSrc; s_slotL[tid] = mSlotL; s_slotR[tid] = mSlotR;
    }
    __syncthreads();
    // ---- level 2: side rows at the batch's own slots; the nodes of the merged rows ----
    for (int idx = tid; idx < K * 2 * K; idx += 1024) {
        const int u = idx / (2 * K), k = idx % (2 * K);
        const int t = k < K ? k : k - K;
        float v = 0.0f;
        if (u < m && t < m) {
            const uint32_t slot = k < K ? s_slotL[t] : s_slotR[t];
            if (slot != UB_NONE) v = a.side[(size_t)u * ld + slot];
        }
        s_side[u][k] = v;
    }
    uint32_t nodeL = UB_NONE, nodeR = UB_NONE;
    int die = UB_INF; // the merge at which the row created by merge `lane` dies again (it is a later merge's R), or INF
    if (wave == 0) {
        if (lane < m) {
            nodeL = a.node_index[mL];
            nodeR = mSrc < 0 ? a.node_index[mR] : UB_NONE;
        }
        for (int w = 0; w < m; ++w) {
            const uint32_t Rw = lane_u32(mR, w);
            if (lane < w && lane < m && mL == Rw && die == UB_INF) die = w;
        }
        if (lane < K) s_die[lane] = die;
    }
    __syncthreads();
    // ---- the cross entries: D[L_t][L_u] for the clusters the batch creates ----
    // A merge whose partner is an old row needs nothing of the table: all those entries at once, one per thread.  A
    // merge whose partner was created in the batch reads the table: those go in merge order (wave 0, lane = u).
    for (int idx = tid; idx < K * K; idx += 1024) {
        const int t = idx / K, u = idx % K;
        if (t < m && u < t && s_die[u] > t && s_src[t] < 0) s_tab[t][u] = ub_average<MODIFIED>(s_side[u][t], s_side[u][K + t]);
    }
    __syncthreads();
    if (wave == 0) {
        for (int t = 1; t < m; ++t) {
            const int src_t = s_src[t];
            if (src_t < 0) continue;
            if (lane < t && die > t) { // the cluster of merge `lane` is alive at merge t
                const int u = lane;
                const float a1 = s_side[u][t]; // D[L_t][L_u] before merge t: the row made by merge u at L_t's slot
                const float a2 = src_t > u ? s_tab[src_t][u] : s_tab[u][src_t];
                s_tab[t][u] = ub_average<MODIFIED>(a1, a2);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();
    // ---- the new rows' minima: 32 lanes per merge over (partials, cross entries), ties by row ----
    {
        const int t = tid >> 5, u = tid & 31;
        float d = UB_BIG;
        uint32_t dj = UB_NONE;
        if (t < m && u < t && s_die[u] > t) ub_take(s_tab[t][u], s_L[u], d, dj);
        if (t < m && u == 31) ub_take(s_pm_d[t], s_pm_j[t], d, dj);
        dpp_min_step<DPP_QUAD_1032, 0xF>(d, dj);
        dpp_min_step<DPP_QUAD_2301, 0xF>(d, dj);
        dpp_min_step<DPP_ROW_HALF_MIRROR, 0xF>(d, dj);
        dpp_min_step<DPP_ROW_MIRROR, 0xF>(d, dj);
        dpp_min_step<DPP_ROW_BCAST15, 0xA>(d, dj);
        if (t < K && u == 31) {
            if (d >= UB_BIG) { d = UB_BIG; dj = UB_NONE; }
            s_min[t] = d;
            s_near[t] = dj;
        }
    }
    __syncthreads();
    if (wave == 0) {
        const float newmin = lane < K ? s_min[lane] : UB_BIG;
        uint32_t newnear = lane < K ? s_near[lane] : UB_NONE;
        // merge t stands iff the reference would have picked L_t: its key beats every row created before it in the
        // batch that is still alive at that point
        bool ok = true;
        for (int s2 = 0; s2 < m; ++s2) {
            const float ms = __uint_as_float(lane_u32(__float_as_uint(newmin), s2));
            const uint32_t Ls = lane_u32(mL, s2);
            const int ds = lane_i32(die, s2);
            if (s2 < lane && ds >= lane) { // (alive when merge `lane` is picked -- also as that merge's own partner)
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
            rec[UB_REC0 + 4 * lane + 0] = __float_as_uint(newmin);
            rec[UB_REC0 + 4 * lane + 1] = newnear;
            rec[UB_REC0 + 4 * lane + 2] = (uint32_t)die;
        }
        if (lane == 0) rec[0] = (uint32_t)V;
        if (lane < V) { // the merged rows' bookkeeping (UPGMA.cpp:268-287) and their slots
            a.left[done + lane] = (int32_t)nodeL;
            a.right[done + lane] = (int32_t)(mSrc < 0 ? nodeR : (uint32_t)n + done + (uint32_t)mSrc);
            const bool dies = die < V;
            const uint32_t new_slot = sb + (uint32_t)lane;
            a.node_index[mL] = dies ? UB_NONE : (uint32_t)n + done + (uint32_t)lane; // node id of the k-th cluster = n + k
            if (mSrc < 0) a.node_index[mR] = UB_NONE;
            a.row_of[mSlotL] = UB_NONE;
            if (mSrc < 0) a.row_of[mSlotR] = UB_NONE;
            a.row_of[new_slot] = dies ? UB_NONE : mL;
            if (!dies) {
                a.slot_of[mL] = new_slot;
                a.min_dist[mL] = newmin;
                a.nearest[mL] = newnear;
            }
        }
        if (lane == 0) {
            st_next[0] = done + (uint32_t)V;
            st_next[1] = ns - (uint32_t)V;
            st_next[2] = err | walk_err;
            st_next[3] = cuts + (V < m ? 1u : 0u);
            st_next[4] = sb + (uint32_t)V;
        }
    }
    // the cross entries for the commit (only the pairs it will write are read there)
    for (int idx = tid; idx < K * K; idx += 1024) rec[UB_REC0 + 4 * K + idx] = __float_as_uint(s_tab[idx / K][idx % K]);
}

// ---- launch 3 of a batch: the commit ----------------------------------------------------------------------------------
// (the bookkeeping of the merged rows -- node_index, row_of, slot_of, min_dist, nearest -- has been written by the resolve
//  kernel: a slot whose row took part in a standing merge already reads row_of = NONE here, the new clusters' slots
//  read their rows)
template <int K, bool MODIFIED>
__global__ __launch_bounds__(256) void upgma_batch_commit_kernel(UpgmaBatchArgs a, int parity)
{
    __shared__ float s_sv[K][256 + 1]; // the side values of my 256 slots, to be written out row by row
    __shared__ uint32_t s_xrow[256];   // the row of each of those slots, NONE where nothing is written
    const int tid = threadIdx.x, b = blockIdx.x, n = a.n;
    const size_t ld = (size_t)a.ld;
    const int lane = tid & 63, wave = tid >> 6;
    const uint32_t gid = (uint32_t)b * 256 + tid; // slot p, row j AND position of the sorted order
    const uint32_t* st = a.state + 8 * parity;
    const uint32_t* rec = a.rec;
    // ---- level 1: lane t of every wave holds merge t's record ----
    const uint32_t ns = st[1], sb = st[4];
    const int V = (int)rec[0];
    uint32_t mL = UB_NONE, mR = UB_NONE, mPosL = UB_NONE, mPosR = UB_NONE, mMinBits = 0u, mNear = UB_NONE;
    int mDie = UB_INF;
    if (lane < K) {
        const uint32_t* h = a.hdr + UB_HDR0 + UB_HDR_STRIDE * lane;
        mL = h[0]; mR = h[1]; mPosL = h[4]; mPosR = h[5];
        mMinBits = rec[UB_REC0 + 4 * lane + 0];
        mNear = rec[UB_REC0 + 4 * lane + 1];
        mDie = (int)rec[UB_REC0 + 4 * lane + 2];
    }
    const bool is_row = gid < (uint32_t)n, is_slot = gid < (uint32_t)a.ld;
    const uint32_t my_node = is_row ? a.node_index[gid] : UB_NONE;
    const uint32_t my_near = is_row ? a.nearest[gid] : UB_NONE;
    const uint32_t x = is_slot ? a.row_of[gid] : UB_NONE; // the row of my slot
    const uint2* cur = parity ? a.sorted1 : a.sorted0;
    uint2* nxt = parity ? a.sorted0 : a.sorted1;
    uint2 e = make_uint2(0xFFFFFFFFu, UB_NONE), ep = make_uint2(0u, 0u);
    if (gid < ns) e = cur[gid];
    if (gid > 0 && gid <= ns) ep = cur[gid - 1];
    float sv[K];
#pragma unroll
    for (int t = 0; t < K; ++t) sv[t] = is_slot ? a.side[(size_t)t * ld + gid] : 0.0f;
    if (V == 0) return; // nothing stands (finished, or an error)
    // ---- commit my slot: the entries of the V new clusters towards my row ----
    bool passes = x != UB_NONE; // my row only passes through the batch (it is in no standing merge)
    bool mine = is_row && my_node != UB_NONE;
    uint32_t near = my_near;
    bool removed = gid == ns;
    uint32_t before = 0; // entries of the order in front of position gid that leave
#pragma unroll
    for (int t = 0; t < K; ++t) {
        if (t < V) {
            const uint32_t Lt = lane_u32(mL, t), Rt = lane_u32(mR, t), pl = lane_u32(mPosL, t), pr = lane_u32(mPosR, t);
            passes = passes && Lt != x && Rt != x;
            mine = mine && Lt != gid && Rt != gid;
            if (near == Rt) near = Lt; // the renames of the batch, in merge order
            removed = removed || pl == gid || pr == gid;
            before += (pl < gid ? 1u : 0u) + (pr != UB_NONE && pr < gid ? 1u : 0u);
        }
    }
    // (only now is `passes` final: the slot of a NEW cluster reads its row from row_of -- the resolve kernel has set it --
    //  and must not write: what the clusters hold towards each other are the cross entries below)
#pragma unroll
    for (int t = 0; t < K; ++t) {
        if (t < V) {
            if (passes) a.D[(size_t)lane_u32(mL, t) * ld + gid] = sv[t]; // the cluster's own row (it keeps its left child's), along the slots
            s_sv[t][tid] = sv[t];
        }
    }
    s_xrow[tid] = passes ? x : UB_NONE;
    __syncthreads();
    {   // my row at the new clusters' slots: V consecutive floats per row -- K lanes write one row's run together
        constexpr int RPI = 64 / K; // rows per store instruction of a wave
        const int tt = lane % K, sub = lane / K;
        const size_t first = (size_t)sb;
#pragma unroll 4
        for (int q = 0; q < 64 / RPI; ++q) {
            const int r = wave * 64 + q * RPI + sub;
            const uint32_t xr = s_xrow[r];
            if (xr != UB_NONE && tt < V) a.D[(size_t)xr * ld + first + tt] = s_sv[tt][r];
        }
    }
    // (slots of rows that take part in a standing merge are dead from here on; what the new clusters hold towards each
    //  other are the cross entries)
    if (b == 0) {
        for (int idx = tid; idx < K * K; idx += 256) {
            const int t = idx / K, u = idx % K;
            if (u < t && t < V && (int)rec[UB_REC0 + 4 * u + 2] > t) {
                const float v = __uint_as_float(rec[UB_REC0 + 4 * K + idx]);
                const uint32_t Lt = a.hdr[UB_HDR0 + UB_HDR_STRIDE * t], Lu = a.hdr[UB_HDR0 + UB_HDR_STRIDE * u];
                a.D[(size_t)Lt * ld + ((size_t)sb + u)] = v;
                a.D[(size_t)Lu * ld + ((size_t)sb + t)] = v;
            }
        }
    }
    // ---- my row's nearest under the renames of the batch ----
    if (mine && near != my_near) a.nearest[gid] = near;
    // ---- the sorted order of the next batch: position gid of the current one (gid == ns: the place behind the end) ----
    if (gid <= ns) {
        uint32_t ins_before = 0;
#pragma unroll
        for (int s2 = 0; s2 < K; ++s2) { // rows the batch created and that are still alive: they enter at their new keys
            if (s2 < V) {
                const int die = lane_i32(mDie, s2);
                const uint32_t kb = lane_u32(mMinBits, s2), row = lane_u32(mL, s2);
                if (die >= V && ub_less(kb, row, e.x, e.y)) { // in front of me (my slot e is +inf at gid == ns)
                    ++ins_before;
                    if (gid == 0 || !ub_less(kb, row, ep.x, ep.y)) { // ... and not in front of my predecessor: I place it
                        uint32_t at = gid - before;
                        for (int s3 = 0; s3 < V; ++s3) {
                            const int die3 = lane_i32(mDie, s3); // (readlane: also from lanes that are not in this branch)
                            const uint32_t kb3 = lane_u32(mMinBits, s3), row3 = lane_u32(mL, s3);
                            if (s3 != s2 && die3 >= V && ub_less(kb3, row3, kb, row)) ++at;
                        }
                        nxt[at] = make_uint2(kb, row);
                        a.pos[row] = at;
                        if (at < (uint32_t)UPGMA_BATCH_CAND) a.cand[at] = make_uint4(kb, row, lane_u32(mNear, s2), 0u);
                    }
                }
            }
        }
        if (!removed) {
            const uint32_t at = gid - before + ins_before;
            nxt[at] = e;
            a.pos[e.y] = at;
            if (at < (uint32_t)UPGMA_BATCH_CAND) {
                uint32_t nr = a.nearest[e.y]; // (a row the batch only passed through: its stored nearest is the old one)
                for (int t = 0; t < V; ++t) {
                    const uint32_t Rt = lane_u32(mR, t), Lt = lane_u32(mL, t);
                    if (nr == Rt) nr = Lt;
                }
                a.cand[at] = make_uint4(e.x, e.y, nr, 0u);
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

// ---- compaction: the live slots to the front (see LAYOUT) -------------------------------------------------------------
// One workgroup: remap[p] = where slot p moves (NONE: dead), row_of packed in place (a slot only moves to a smaller
// index, and a chunk is read completely before any of it is written), slot_of of the live rows, the next free slot.
__global__ __launch_bounds__(1024) void upgma_compact_map_kernel(UpgmaBatchArgs a, int parity, uint32_t used)
{
    __shared__ uint32_t s_cnt[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    uint32_t base = 0;
    for (uint32_t c0 = 0; c0 < used; c0 += 1024) {
        const uint32_t p = c0 + (uint32_t)tid;
        const uint32_t row = p < used ? a.row_of[p] : UB_NONE;
        const bool live = row != UB_NONE;
        const uint64_t mask = __ballot(live);
        if (lane == 0) s_cnt[wave] = (uint32_t)__popcll(mask);
        __syncthreads(); // (every row_of of the chunk has been read)
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const uint32_t c = s_cnt[w];
            before += w < wave ? c : 0u;
            total += c;
        }
        const uint32_t np = base + before + (uint32_t)__popcll(mask & lt_mask);
        if (p < used) a.remap[p] = live ? np : UB_NONE;
        if (live) {
            a.row_of[np] = row;
            a.slot_of[row] = np;
        }
        base += total;
        __syncthreads(); // (s_cnt is read; the next chunk may overwrite it)
    }
    for (uint32_t p = base + (uint32_t)tid; p < used; p += 1024) a.row_of[p] = UB_NONE;
    if (tid == 0) a.state[8 * parity + 4] = base;
}

// One workgroup per row: a live row's entries move to their slots' new places, 2048 slots at a time.
__global__ __launch_bounds__(256) void upgma_compact_rows_kernel(UpgmaBatchArgs a, uint32_t used)
{
    constexpr int PER = 8;
    const uint32_t x = blockIdx.x;
    const int tid = threadIdx.x;
    const uint32_t sx = a.slot_of[x]; // (already the new slot of a live row; anything of a dead one)
    if (sx >= (uint32_t)a.ld || a.row_of[sx] != x) return;
    float* row = a.D + (size_t)x * (size_t)a.ld;
    for (uint32_t c0 = 0; c0 < used; c0 += 256 * PER) {
        float v[PER];
        uint32_t np[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const uint32_t p = c0 + (uint32_t)(u * 256 + tid);
            np[u] = p < used ? a.remap[p] : UB_NONE;
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const uint32_t p = c0 + (uint32_t)(u * 256 + tid);
            v[u] = np[u] != UB_NONE ? row[p] : 0.0f;
        }
        __syncthreads(); // the chunk is in registers: its values may now land on places of the same chunk (never beyond it)
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if (np[u] != UB_NONE) row[np[u]] = v[u];
    }
}

hipError_t launch_upgma_compact(const UpgmaBatchArgs& a, int parity, long long slots_used, hipStream_t stream)
{
    const uint32_t used = (uint32_t)std::min<long long>(slots_used, a.ld);
    hipLaunchKernelGGL(upgma_compact_map_kernel, dim3(1), dim3(1024), 0, stream, a, parity, used);
    hipLaunchKernelGGL(upgma_compact_rows_kernel, dim3((unsigned)a.n), dim3(256), 0, stream, a, used);
    return hipGetLastError();
}

// `count` batches (three launches each), the first of them batch number `first` (its parity selects the state buffers).
// Before batch i at most slots_used + (i - first) * k slots are in use: its launches cover those.
hipError_t launch_upgma_batches(const UpgmaBatchArgs& a, bool modified, int k, int first, int count, long long slots_used, hipStream_t stream)
{
    const dim3 block(256);
    for (int i = first; i < first + count; ++i) {
        const int parity = i & 1;
        const long long slots = std::min<long long>(a.ld, slots_used + (long long)(i - first) * k);
        const int g = (int)std::min<long long>((slots + 255) / 256, a.n_blocks);
        const dim3 grid_rows((unsigned)g), grid_commit((unsigned)std::max<long long>(g, a.n / 256 + 1));
#define UB_LAUNCH(KK)                                                                                                     \
    do {                                                                                                                  \
        if (modified) {                                                                                                   \
            hipLaunchKernelGGL((upgma_batch_rows_kernel<KK, true>), grid_rows, block, 0, stream, a, parity);              \
            hipLaunchKernelGGL((upgma_batch_resolve_kernel<KK, true>), dim3(1), dim3(1024), 0, stream, a, parity, g);     \
            hipLaunchKernelGGL((upgma_batch_commit_kernel<KK, true>), grid_commit, block, 0, stream, a, parity);          \
        } else {                                                                                                          \
            hipLaunchKernelGGL((upgma_batch_rows_kernel<KK, false>), grid_rows, block, 0, stream, a, parity);             \
            hipLaunchKernelGGL((upgma_batch_resolve_kernel<KK, false>), dim3(1), dim3(1024), 0, stream, a, parity, g);    \
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
