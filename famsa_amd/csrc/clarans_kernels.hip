// clarans_kernels.hip -- CLARANS k-medoids of the MedoidTree heuristic on the device.
//
// What it reproduces: CLARANS::operator() (reference tree/Clustering.cpp:17-305) as called by
// FastTree::clusterSeeds (tree/FastTree.cpp:366-436) on the float distance triangle of the sample.
// The search is a chain of "steps": draw a non-medoid position xx, evaluate for every medoid slot
// k the cost change of replacing medoid k by candidate[xx], accept the best k if it lowers the
// cost, and stop after `corrected` steps without an accept.  ONE WORKGROUP RUNS ONE LOCAL SEARCH from its first step to
// its last (clarans_search_kernel): every position's state in its registers, deltas[slot] accumulated over the
// non-medoids in ascending position -- the reference's float additions in the reference's order; ties, comparison
// directions and float operation order follow the reference line by line; the running cost is summed sequentially from
// each accept's addends.  The searches of several host threads share a launch, one workgroup each, for a time slice; the
// host reads the `done` flags between slices.
//
// Layout: D = the sample members' full symmetric float matrix (D[i*n + j]).  All search state is
// kept per candidate POSITION (not per member), so the lanes of a wave read it coalesced:
// st[pos] = nearest / second-nearest bookkeeping, DMt[mm*n + pos] = distance of the member at pos to
// the medoid in slot mm (kept in step with the swaps so that the reference's updateAssignment scan
// is k coalesced loads instead of k scattered triangle entries per lane).
//
// Shapes: 1 <= n - k <= 2048 non-medoids (every position's state in the registers of one workgroup), k <= 1024; the
// host side answers LCSGPU_E_UNSUPPORTED beyond that and the caller searches on the host.  (Rounds 1-5 also carried:
// a launch per round with a workgroup per pending step -- all steps up to the next accept evaluated at once from the
// same state --, its two-launch form, an evaluation with per-slot lists and a one-XCD persistent kernel; each was
// measured slower than what replaced it -- profiles/clarans_rounds_r03.txt, clarans_rounds_r04.txt,
// clarans_lists_r04.txt, c5_search_r05.txt, CHANGELOG.md -- and removed.)
#include <hip/hip_runtime.h>

#include <cfloat>
#include <algorithm>
#include <type_traits>
#include <climits>
#include <cstdint>

#include "lcs_kernels.h"
#include "fasttree_kernels.h"

#ifndef CLARANS_QC
#define CLARANS_QC 8 // (4: 27.3 ms for a chain of 2000 members / 100 medoids, 8: 25.9, 16: 31.0 -- profiles/qc_r06.txt)
#endif

namespace lcsgpu {

namespace {

// D is the sample's FULL symmetric matrix, D[i * n + j]: the distances of one member to all others are one contiguous row
// (n x 4 B = 8 KB at 2000 members), so the per-step gathers D[x][member at position ..] use every byte of the 32-byte
// sectors they touch.  (In the packed triangle the part j > i of that "row" is a column walk, one sector per element:
// 34 KB instead of 8 KB per gathered row, and the rounds of ~20 concurrent searches -- 16 step workgroups each, two
// such rows per workgroup and round -- are bound by exactly that traffic)
__device__ __forceinline__ size_t sq_at(int n, int i, int j) { return (size_t)i * (size_t)n + (size_t)j; }

// wave-level reductions by DPP (see tree_kernels.hip, wave_first_min): the moves between lanes stay inside the VALU
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false); }
// (value, index) with "a valid index beats none, then the smaller value, then the smaller index"; result wave-uniform
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void dpp_first_min_step(float& v, int& i)
{
    const float ov = __int_as_float(dpp_i32<CTRL, ROW_MASK>(__float_as_int(v)));
    const int oi = dpp_i32<CTRL, ROW_MASK>(i);
    if (oi != INT_MAX && (i == INT_MAX || ov < v || (ov == v && oi < i))) { v = ov; i = oi; }
}
__device__ __forceinline__ void wave_first_min_valid(float& v, int& i)
{
    dpp_first_min_step<0xB1, 0xF>(v, i);  // quad_perm [1,0,3,2]
    dpp_first_min_step<0x4E, 0xF>(v, i);  // quad_perm [2,3,0,1]
    dpp_first_min_step<0x141, 0xF>(v, i); // row_half_mirror
    dpp_first_min_step<0x140, 0xF>(v, i); // row_mirror
    dpp_first_min_step<0x142, 0xA>(v, i); // row_bcast:15
    dpp_first_min_step<0x143, 0xC>(v, i); // row_bcast:31
    v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
    i = __builtin_amdgcn_readlane(i, 63);
}

// the search's state block (16 words; the host reads a copy after every launch)
enum { ST_P = 0 /* next draw */, ST_DONE = 1, ST_ROUNDS = 3 /* accepts */, ST_FRESH = 4 /* 1 = nothing has run yet */, ST_COST = 5, ST_ERR = 6,
       ST_MORE_DRAWS = 7 /* stopped for want of pre-drawn positions */, ST_OFF = 8 /* steps of the window already evaluated */,
       ST_FIRST = 10 /* no accept yet in this search */,
       // statistics of the search (LCSGPU_PROFILE): groups of steps, steps looked at, steps up to the accepted one, steps that
       // ended without a walk because no member was closer to the candidate than to its medoid / because no slot could go negative
       ST_N_ROUNDS = 11, ST_N_STEPS = 12, ST_N_USEFUL = 13, ST_N_NOB = 14, ST_N_NOP = 15,
       // the chain (clarans_chain_kernel): the local search it is in, the cheapest search's cost so far, whether the search it is in has to be started
       ST_ITER = 16, ST_BEST = 17, ST_NEED_INIT = 18,
       ST_TICKS = 19 /* 100 MHz ticks the chain's workgroup has run (LCSGPU_PROFILE) */, ST_CU = 20 /* where its last launch ran: XCC id << 8 | CU id */ };
// why an evaluation ended
enum { WHY_WALKED = 0, WHY_NO_B = 1, WHY_NO_P = 2 };

// the first window of a local search has `corrected` steps, the later ones corrected - 1 (the reference resets its step
// counter to 1 after an accept, Clustering.cpp:82-247)
__device__ __forceinline__ int window_size(int corrected, int first) { return first ? corrected : (corrected > 0 ? corrected - 1 : 0); }

} // namespace

// Per-position state: st[pos] = {distance to the nearest medoid, to the second nearest, slot of the
// nearest, slot of the second} of the member at candidate position pos (only positions >= n_medoids
// are ever read; the reference's bookkeeping for the medoids themselves is write-only).
__device__ __forceinline__ float4 pack_state(float dn, float ds, int an, int as)
{
    return make_float4(dn, ds, __int_as_float(an), __int_as_float(as));
}

// CLARANS::updateAssignment (Clustering.cpp:262-305) folded over slots in ascending order
struct Nearest2 {
    float dn = FLT_MAX, ds = FLT_MAX;
    int an = -1, as = -1;
    __device__ __forceinline__ void feed(float d, int mm)
    {
        if (d < dn) { ds = dn; as = an; dn = d; an = mm; }
        else if (d < ds) { ds = d; as = mm; }
    }
};

// Start of one local search (Clustering.cpp:49-79), by the search's own workgroup: every non-medoid's distances to the
// medoid slots (DMt[mm * n + pos]) and assignment, the addends of the initial cost in position order.  The candidate
// order has just been written by this workgroup: it is read with workgroup-scope loads (never through the scalar cache).
__device__ __forceinline__ void chain_init(const ClaransArgs& a)
{
    constexpr int PER = 4;
    const int k = a.n_medoids, n = a.n_elems, tid = threadIdx.x;
    int y[PER];
    Nearest2 nb[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int pos = k + tid + 512 * u;
        y[u] = pos < n ? __hip_atomic_load(a.cand + pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0;
    }
    for (int mm = 0; mm < k; ++mm) {
        const int med = __builtin_amdgcn_readfirstlane(__hip_atomic_load(a.cand + mm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        const float* row = a.D + (size_t)med * (size_t)n;
        float d[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) d[u] = row[y[u]];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int pos = k + tid + 512 * u;
            if (pos < n) {
                a.DMt[(size_t)mm * n + pos] = d[u];
                nb[u].feed(d[u], mm);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int pos = k + tid + 512 * u;
        if (pos >= n) continue;
        a.st[pos] = pack_state(nb[u].dn, nb[u].ds, nb[u].an, nb[u].as);
        a.cost_log[pos - k] = nb[u].dn;
    }
}

// running cost: c += addend for every logged addend, in order; zeros are the identity (c starts at +0.0f and can
// never become -0.0f), so only the others are walked.  Valid in thread 0.
__device__ __forceinline__ float cost_accumulate(const float* cost_log, int len, float c, float* s_f, float* s_nz)
{
    constexpr int CH = 2048, PER = CH / 512;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    for (int c0 = 0; c0 < len; c0 += CH) {
        const int cnt = min(CH, len - c0);
        float v[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) { // one trip to memory for the whole pass
            const int t = tid + 512 * u;
            v[u] = t < cnt ? cost_log[c0 + t] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) s_f[tid + 512 * u] = v[u];
        __syncthreads();
        if (wave == 0) {
            int m = 0;
            for (int base = 0; base < cnt; base += 256) {
                float g[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) g[u] = s_f[base + 64 * u + lane]; // zero beyond cnt
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint64_t mask = __ballot(g[u] != 0.0f);
                    if (g[u] != 0.0f) s_nz[m + __popcll(mask & lt_mask)] = g[u];
                    m += __popcll(mask);
                }
            }
            __builtin_amdgcn_wave_barrier(); // same wave: its LDS writes are performed before its later reads
            if (lane == 0) {
                int t = 0;
                for (; t + 8 <= m; t += 8) {
                    float g[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) g[u] = s_nz[t + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) c = __fadd_rn(c, g[u]);
                }
                for (; t < m; ++t) c = __fadd_rn(c, s_nz[t]);
            }
        }
        __syncthreads();
    }
    return c;
}

// deltas[slot] of the member x drawn at position xx (Clustering.cpp:93-118) and their first minimum over the free slots
// (cpp:121-122) -- as far as the caller can tell the difference: it only asks whether the minimum is negative and, if
// so, which slot has it.  Results are valid in thread 0.
//
// Every slot's delta is a sequential float sum over the non-medoids in position order.  A non-medoid y adds
//   (b) to EVERY slot the same y_j = d(x, y) - dn(y) < 0 when the candidate is closer to it than its medoid (its own slot
//       gets min(d, ds) - dn, which is the same value then), and otherwise
//   (a) to its own slot only x_i = min(d(x, y), ds(y)) - dn(y) >= 0.
// WHICH SLOTS CAN GO NEGATIVE (round 5).  With Y = the sum of the (b) addends and X_s = the sum of slot s's (a) addends,
// the exact value of delta[s] is T = X_s + Y, and the float sum computed in ANY order differs from T by at most
// gamma * (X_s + |Y|), gamma = m u / (1 - m u) < 1.3e-4 for m <= 2049 terms (u = 2^-24: the bound of recursive
// summation; the terms themselves are the reference's float values).  X_s and Y, summed here in whatever order the
// atomics arrive, are sums of equal-signed terms and carry the same relative bound.  So X_s > 1.002 |Y| -- four times
// the margin the two bounds need -- means delta[s] > 0 in the reference's own order of additions: slot s is neither
// the accepted slot nor able to change whether the step is accepted.  Only the other slots -- the set P -- are summed
// exactly (the reference's additions in the reference's order, as before); a step without a (b) entry, or with an empty
// P, cannot be accepted and ends here.  On a 2000-member family sample with 100 medoids 76 % of the steps have no (b)
// entry, 89.5 % an empty P, and the walks of the rest touch 1.8 % of the entries (scripts/clarans_prune_stats.cpp,
// profiles/clarans_prune_r05.txt) -- which is what the LDS port of a CU, shared by the evaluations resident on it, was
// busy with.
//
// The exact sums: each of the 8 waves -- wave w owns the slots [w * kpw, (w + 1) * kpw) -- compacts, in order, the
// entries that can change one of ITS slots in P and walks only those: skipped entries would add +0.0f, the identity.
template <int KPT>
__device__ __forceinline__ void evaluate_step(const ClaransArgs& a, int xx, int x, const int* y_pre, const float4* s_pre, float4* s_e,
                                              float4 (*s_we)[128], float* s_x, float& best_out, int& bk_out, int& why_out)
{
    constexpr int PER = 4, HALF = 1024, SUB = 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = a.n_medoids, n = a.n_elems;
    const int cnt = n - k; // <= 2048: one chunk
    const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    const int kpw = (k + 7) >> 3;
    const int klo = wave * kpw, khi = min(k, klo + kpw);
    // entries: (addend for the own slot, addend for the other slots, own slot)
    float dxy[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) { // the gathers from the triangle, all in flight together
        const int t = tid + 512 * u;
        dxy[u] = (t < cnt && k + t != xx) ? a.D[sq_at(n, x, y_pre[u])] : 0.0f;
    }
    for (int s = tid; s < k; s += 512) s_x[s] = 0.0f;
    if (tid == 0) s_x[CLARANS_MAX_MEDOIDS] = 0.0f;
    float4 ent[PER];
    bool any_b = false;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int t = tid + 512 * u;
        ent[u] = make_float4(0.0f, 0.0f, __int_as_float(-1), 0.0f); // position xx: contributes nothing
        if (t < cnt && k + t != xx) {
            const float dn = s_pre[u].x, ds = s_pre[u].y;
            const float m = ds < dxy[u] ? ds : dxy[u]; // std::min(dxy, ds)
            const float change = __fsub_rn(dxy[u], dn);
            ent[u].x = __fsub_rn(m, dn);                     // goes to deltas[nearest(y)]
            ent[u].y = change < 0.0f ? change : 0.0f;        // goes to every other slot when negative
            ent[u].z = s_pre[u].z;
            any_b |= change < 0.0f;
        }
    }
    if (!__syncthreads_or(any_b ? 1 : 0)) { // (also: the zeros of s_x are in place)
        if (tid == 0) { best_out = 0.0f; bk_out = INT_MAX; why_out = WHY_NO_B; }
        return;
    }
    // X_s and Y, in any order
    float ysum = 0.0f;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int nn = __float_as_int(ent[u].z);
        if (nn < 0) continue;
        if (ent[u].y < 0.0f) ysum += ent[u].y;
        else atomicAdd(&s_x[nn], ent[u].x);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ysum += __shfl_xor(ysum, o);
    if (lane == 0 && ysum != 0.0f) atomicAdd(&s_x[CLARANS_MAX_MEDOIDS], ysum);
    __syncthreads();
    const float bound = -s_x[CLARANS_MAX_MEDOIDS] * 1.002f; // (X_s is read and replaced by its flag by one thread only)
    bool any_p = false;
    for (int s = tid; s < k; s += 512) {
        const bool in_p = s >= a.n_fixed && !(s_x[s] > bound);
        s_x[s] = in_p ? 1.0f : 0.0f;
        any_p |= in_p;
    }
    if (!__syncthreads_or(any_p ? 1 : 0)) {
        if (tid == 0) { best_out = 0.0f; bk_out = INT_MAX; why_out = WHY_NO_P; }
        return;
    }
    // the (a) entries of the slots outside P take no part
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int nn = __float_as_int(ent[u].z);
        if (nn >= 0 && !(ent[u].y < 0.0f) && s_x[nn] == 0.0f) ent[u] = make_float4(0.0f, 0.0f, __int_as_float(-1), 0.0f);
    }
    float acc[KPT];
    int slot[KPT];
#pragma unroll
    for (int q = 0; q < KPT; ++q) {
        acc[q] = 0.0f;
        slot[q] = klo + lane + 64 * q;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int hcnt = min(HALF, cnt - h * HALF);
        if (hcnt <= 0) break;
#pragma unroll
        for (int u = 0; u < PER / 2; ++u) {
            const int t = tid + 512 * u; // position inside this half
            if (t < hcnt) s_e[t] = ent[h * (PER / 2) + u];
        }
        __syncthreads();
        for (int s0 = 0; s0 < hcnt; s0 += SUB) {
            float4 e[SUB / 64];
#pragma unroll
            for (int u = 0; u < SUB / 64; ++u) {
                const int t = s0 + 64 * u + lane;
                e[u] = t < hcnt ? s_e[t] : make_float4(0.0f, 0.0f, __int_as_float(-1), 0.0f);
            }
            int m = 0;
#pragma unroll
            for (int u = 0; u < SUB / 64; ++u) {
                const int nn = __float_as_int(e[u].z);
                const bool mine = e[u].y < 0.0f || (nn >= klo && nn < khi);
                const uint64_t mask = __ballot(mine);
                if (mine) s_we[wave][m + __popcll(mask & lt_mask)] = e[u];
                m += __popcll(mask);
            }
            __builtin_amdgcn_wave_barrier(); // same wave: its LDS writes are performed before its later reads
            int i = 0;
            for (; i + 8 <= m; i += 8) {
                float4 f[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) f[u] = s_we[wave][i + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int nn = __float_as_int(f[u].z);
#pragma unroll
                    for (int q = 0; q < KPT; ++q) acc[q] = __fadd_rn(acc[q], slot[q] == nn ? f[u].x : f[u].y);
                }
            }
            for (; i < m; ++i) {
                const float4 f = s_we[wave][i];
                const int nn = __float_as_int(f.z);
#pragma unroll
                for (int q = 0; q < KPT; ++q) acc[q] = __fadd_rn(acc[q], slot[q] == nn ? f.x : f.y);
            }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
    }
    // std::min_element over the slots of P (a subset of [n_fixed, k)): smallest value, earliest slot among equals.  (A slot
    // outside P holds a positive delta in the reference -- it could only be the minimum of a step that is not accepted.)
    float* s_v = reinterpret_cast<float*>(&s_we[0][0]);
    int* s_k = reinterpret_cast<int*>(&s_we[1][0]);
    float best = 0.0f;
    int bk = INT_MAX;
#pragma unroll
    for (int q = 0; q < KPT; ++q)
        if (slot[q] < khi && s_x[slot[q]] != 0.0f && (bk == INT_MAX || acc[q] < best)) { best = acc[q]; bk = slot[q]; }
    wave_first_min_valid(best, bk); // the wave's first minimum, then the 8 waves' through LDS: one barrier instead of ten
    if (lane == 0) {
        s_v[wave] = best;
        s_k[wave] = bk;
    }
    __syncthreads();
    if (tid == 0) {
        float v = s_v[0];
        int kk = s_k[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) {
            const float v2 = s_v[w];
            const int k2 = s_k[w];
            if (k2 != INT_MAX && (kk == INT_MAX || v2 < v || (v2 == v && k2 < kk))) { v = v2; kk = k2; }
        }
        best_out = v;
        bk_out = kk;
        why_out = WHY_WALKED;
    }
}

// What an accept -- the member x_acc becomes the medoid of slot mm_new -- does to a non-medoid position that keeps its
// member (Clustering.cpp:124-238, branch by branch): dnw = the member's distance to the new medoid; the position's addend
// to the running cost; `need`: the position has to look at all k slots again.
__device__ __forceinline__ void apply_accept_to_position(float4& st, float dnw, int mm_new, float& addend, bool& need)
{
    const float4 s0 = st;
    const float dn_y = s0.x, ds_y = s0.y;
    const int an_y = __float_as_int(s0.z), as_y = __float_as_int(s0.w);
    float4 out = s0;
    if (an_y == mm_new) { // its medoid is the one that left
        if (dnw < ds_y) {
            out.x = dnw;
            addend = __fsub_rn(dnw, dn_y);
        } else {
            need = true;
            addend = __fsub_rn(ds_y, dn_y);
        }
    } else if (dnw < dn_y) {
        out = pack_state(dnw, dn_y, mm_new, an_y);
        addend = __fsub_rn(dnw, dn_y);
    } else if (as_y != mm_new && dnw < ds_y) {
        out.y = dnw;
        out.w = __int_as_float(mm_new);
    } else if (as_y != mm_new && dnw > ds_y) {
        // Neither of its two nearest slots is the one that changed, and the new medoid is strictly farther than the
        // second: the reference rescans here (Clustering.cpp:228-232) and arrives at the same two distances.  Which
        // SLOTS it names can differ from what is kept here only among slots at EQUAL distance (the reference's
        // incremental branches themselves leave such states: d_new == dn with a smaller slot keeps `an`; a later
        // rescan swaps them) -- and with dn == ds either order gives the same addends (own = other = 0 or both
        // d - dn), the same cost and the same later branches' values; only the labels of a tie may differ.
        // Equality with the second stays with the rescan: there the slot order decides what is stored.
    } else {
        need = true;
    }
    st = out;
}

template <int PER>
__device__ __forceinline__ void rescan_positions(const ClaransArgs& a, const bool* need, const float* d_new, float4* s_pre, int mm_new)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = a.n_medoids, n = a.n_elems;
    // The rescans (a few dozen positions per accept, but nearly every wave has one): a lane walking its position's k
    // distances alone waits for k / 8 dependent batches of scattered loads.  Instead the WAVE takes each such position:
    // lane m loads the distance to slot m (+ 64, ...), all of a group's loads in flight together, and the two nearest
    // slots are two wave minima -- "first minimum over the slots, then first minimum over the rest", which is what
    // the sequential scan of Clustering.cpp:262-305 arrives at.  (Slice by slice: taking a wave's positions of all four
    // slices four at a time -- fewer rounds -- made a chain SLOWER, 27.9 against 25.9 ms: the kernel sits at its 128
    // registers and the shared loop spilled more; round 6, second session.)
    const int kq = (k + 63) >> 6;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        unsigned long long todo = __ballot(need[u]);
        while (todo) {
            int rl[4];
            int cnt = 0;
            while (cnt < 4 && todo) {
                rl[cnt++] = (int)__builtin_ctzll(todo);
                todo &= todo - 1;
            }
            if (kq <= 2) {
                float v[4][2];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int pos_r = k + wave * 64 + (c < cnt ? rl[c] : rl[0]) + 512 * u;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int mm = lane + 64 * q;
                        v[c][q] = (c < cnt && mm < k) ? a.DMt[(size_t)mm * n + pos_r] : FLT_MAX;
                    }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c >= cnt) break;
                    const float dnw_r = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d_new[u]), rl[c]));
                    float dv[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) dv[q] = (lane + 64 * q == mm_new) ? dnw_r : v[c][q];
                    float v1 = FLT_MAX, v2 = FLT_MAX;
                    int i1 = INT_MAX, i2 = INT_MAX;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int mm = lane + 64 * q;
                        if (mm < k && (dv[q] < v1 || i1 == INT_MAX)) { v1 = dv[q]; i1 = mm; }
                    }
                    wave_first_min_valid(v1, i1);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int mm = lane + 64 * q;
                        if (mm < k && mm != i1 && (dv[q] < v2 || i2 == INT_MAX)) { v2 = dv[q]; i2 = mm; }
                    }
                    wave_first_min_valid(v2, i2);
                    const bool has1 = i1 != INT_MAX && v1 < FLT_MAX, has2 = i2 != INT_MAX && v2 < FLT_MAX;
                    if (lane == rl[c]) s_pre[u] = pack_state(has1 ? v1 : FLT_MAX, has2 ? v2 : FLT_MAX, has1 ? i1 : -1, has2 ? i2 : -1);
                }
            } else { // more than 128 slots: every such lane scans its own column (the apply kernel's loop)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c < cnt && lane == rl[c]) {
                        const int pos = k + tid + 512 * u;
                        const float dnw = d_new[u];
                        Nearest2 nb;
                        const float* col = a.DMt + pos;
                        for (int m0 = 0; m0 < k; m0 += 8) {
                            float vv[8];
#pragma unroll
                            for (int q = 0; q < 8; ++q) vv[q] = col[(size_t)min(m0 + q, k - 1) * n];
#pragma unroll
                            for (int q = 0; q < 8; ++q)
                                if (m0 + q < k) nb.feed(m0 + q == mm_new ? dnw : vv[q], m0 + q);
                        }
                        s_pre[u] = pack_state(nb.dn, nb.ds, nb.an, nb.as);
                    }
                }
            }
        }
    }
}

// what a search keeps between launches (the state block) -- in registers while a workgroup runs: a chain's second search
// must not read back through the scalar cache what the first one has just stored
struct SearchRegs {
    int P, off, first, accepts, fresh;
    float cost; // (kept by thread 0)
    int n_groups, n_steps, n_useful, n_nob, n_nop;
};

// ---------------------------------------------------------------------------------------------------------------
// A WHOLE LOCAL SEARCH IN ONE WORKGROUP (round 5).
//
// Since only the slots that can go negative are summed (evaluate_step), 85 % of the steps end after their loads and a
// flag: a step is mostly latency, and what the launch-per-round form paid per accept -- a kernel boundary, four dependent
// load levels from a cold cache, 17 workgroups per search -- was more than the work between two accepts.  Here one
// workgroup runs the reference's loop as it stands (Clustering.cpp:82-247: draw, evaluate, accept or count, stop after
// `corrected` steps without an accept), the state of every position in its registers from the first step to the last:
//   * the next Q = 16 pending steps' rows of D are read (the draws do not depend on the state) and reduced to one bit
//     per step, "some member is closer to the candidate than to its medoid" -- asked member by member, so a row is read as
//     it lies in memory; one barrier for the group;
//   * only the flagged steps are evaluated, in order, by evaluate_step (their rows come from the cache now); the first
//     negative minimum is the accept, the rest of the group is dropped;
//   * the accept is applied to the registers (the branches of Clustering.cpp:124-238), the running cost takes the
//     accept's addends in order, and the loop goes on with the next draw.
// No other workgroup reads what this one writes: nothing to wait for, nothing that can deadlock.  128 VGPRs and 37 KB of
// LDS, so that the workgroup finds room on a CU that also runs workgroups of the LCS kernels -- with 240 VGPRs / 62 KB it
// waited for a CU to drain and the stage was no faster than with the rounds (profiles/c5_search_r05.txt).  A launch ends for a search when it is done, when the pre-drawn positions run out
// (ST_MORE_DRAWS: the host draws more) or when its time slice is over; the state is where the next launch finds it.
// Returns 1: the local search is over, 2: out of pre-drawn positions, 0: the time slice is over.
template <int KPT>
__device__ __forceinline__ int clarans_search_body(const ClaransArgs& a, SearchRegs& R, long long slice_ticks, long long t_begin)
{
    constexpr int PER = 4, Q = 16, MEMBERS = CLARANS_MAX_MEDOIDS + CLARANS_MAX_NONMEDOIDS;
    __shared__ float4 s_e[1024];        // 16 KB   evaluate_step's staging
    __shared__ float4 s_we[8][128];     // 16 KB
    __shared__ float4 s_xx_state;       // the rebuilt state of the position that received the replaced medoid
    __shared__ float s_x[CLARANS_MAX_MEDOIDS + 8];
    // by MEMBER: its distance to its medoid (0 for a medoid) -- how a position's owner hands the value to the thread that
    // holds the member in the flags phase; lives in the staging between an accept and the next evaluation
    float* s_dn = reinterpret_cast<float*>(&s_we[0][0]);
    static_assert(MEMBERS * sizeof(float) <= sizeof(s_we), "s_dn");
    // the candidate order stays in memory (a.cand, kept current): read with workgroup-scope loads, never through the scalar cache
    auto cand_at = [&](int i) { return __hip_atomic_load(a.cand + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto cand_at_uniform = [&](int i) { return __builtin_amdgcn_readfirstlane(cand_at(i)); }; // (i wave-uniform: the row's base in scalar registers)
    __shared__ unsigned s_flag[2][9];   // per wave: the steps of the group with a (b) entry, by group parity; [8]: time is up
    __shared__ int s_res[4];            // an evaluation's result for everybody
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = a.n_medoids, n = a.n_elems, corrected = a.corrected, cnt = n - k;
    int P = R.P, off = R.off, first = R.first, accepts = R.accepts;
    const int fresh = R.fresh;
    float cost = R.cost;
    int n_groups = R.n_groups, n_steps = R.n_steps, n_useful = R.n_useful, n_nob = R.n_nob, n_nop = R.n_nop;
    int y_pre[PER];
    float4 s_pre[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int pos = k + tid + 512 * u;
        y_pre[u] = pos < n ? cand_at(pos) : 0;
        s_pre[u] = pos < n ? a.st[pos] : make_float4(0.f, 0.f, 0.f, 0.f);
        if (pos < n) s_dn[y_pre[u]] = s_pre[u].x;
    }
    for (int mm = tid; mm < k; mm += 512) s_dn[cand_at(mm)] = 0.0f;
    __syncthreads();
    float dnm[MEMBERS / 512]; // my members' (tid, tid + 512, ...) distances to their medoids; beyond n: never flagged
#pragma unroll
    for (int j = 0; j < MEMBERS / 512; ++j) dnm[j] = tid + 512 * j < n ? s_dn[tid + 512 * j] : -FLT_MAX;
    if (fresh) cost = cost_accumulate(a.cost_log, cnt, cost, reinterpret_cast<float*>(s_e), reinterpret_cast<float*>(s_we)); // the initial cost (Clustering.cpp:49-79)
    int status = 0; // 1: the search is over   2: out of pre-drawn positions
    int expired = 0;
    for (unsigned g = 0;; ++g) {
        const int W = window_size(corrected, first);
        if (off >= W) { // `corrected` steps (corrected - 1 after an accept) without an accept
            P += W;
            off = 0;
            status = 1;
            break;
        }
        const int avail = a.draws_len - (P + off);
        if (avail <= 0) {
            status = 2;
            break;
        }
        if (expired) break; // this look's share of the time is used up (one reading of the clock for the whole workgroup, below)
        const int qn = min(Q, min(W - off, avail));
        const int32_t* dr = a.draws + P + off;
        // ---- the group's flags: all rows in flight together.  "Some non-medoid other than the candidate is closer to the
        // candidate than to its medoid" does not depend on the order of the positions, so it is asked member by member:
        // a row of D is read as it lies in memory (a wave's load = 2 cache lines; gathered by position it is 64, and the
        // L1's one tag look-up per clock made that 12 of a group's 15 us). ----
        unsigned m = 0;
        {
            const int jn = (n + 511) >> 9;
            auto flags_of = [&](auto JN) { // (unconditional loads: four rows are requested before the first is looked at)
                constexpr int J = decltype(JN)::value, QC = CLARANS_QC; // rows in flight per batch of the flags phase
#pragma unroll
                for (int s0 = 0; s0 < Q; s0 += QC) {
                    if (s0 >= qn) break;
                    int x[QC];
                    float d[QC][J];
#pragma unroll
                    for (int c = 0; c < QC; ++c) x[c] = cand_at_uniform(dr[min(s0 + c, qn - 1)]);
#pragma unroll
                    for (int c = 0; c < QC; ++c) {
                        const float* row = a.D + (size_t)x[c] * (size_t)n;
#pragma unroll
                        for (int j = 0; j < J; ++j) d[c][j] = row[min(tid + 512 * j, n - 1)];
                    }
#pragma unroll
                    for (int c = 0; c < QC; ++c)
#pragma unroll
                        for (int j = 0; j < J; ++j)
                            if (tid + 512 * j != x[c] && __fsub_rn(d[c][j], dnm[j]) < 0.0f) m |= 1u << (s0 + c);
                }
            };
            if (jn <= 4) flags_of(std::integral_constant<int, 4>());
            else flags_of(std::integral_constant<int, MEMBERS / 512>());
        }
        unsigned wm = 0;
#pragma unroll
        for (int s = 0; s < Q; ++s)
            if (__ballot((m >> s) & 1u)) wm |= 1u << s;
        if (lane == 0) s_flag[g & 1][wave] = wm;
        if (tid == 0) s_flag[g & 1][8] = wall_clock64() - t_begin > slice_ticks;
        __syncthreads();
        expired = __builtin_amdgcn_readfirstlane((int)s_flag[g & 1][8]);
        unsigned flags = 0;
#pragma unroll
        for (int w = 0; w < 8; ++w) flags |= s_flag[g & 1][w];
        flags = (unsigned)__builtin_amdgcn_readfirstlane((int)flags) & ((1u << qn) - 1u);
        // ---- the flagged steps, in order, up to the first accept ----
        int acc_s = -1, mm_new = 0;
        for (unsigned todo = flags; todo; todo &= todo - 1) {
            const int s = __builtin_ctz(todo);
            const int xx = dr[s];
            float best = 0.0f;
            int bk = INT_MAX, why = WHY_WALKED;
            evaluate_step<KPT>(a, xx, cand_at_uniform(xx), y_pre, s_pre, s_e, s_we, s_x, best, bk, why);
            if (tid == 0) {
                s_res[0] = __float_as_int(best);
                s_res[1] = bk;
                s_res[2] = why;
            }
            __syncthreads();
            const float r_delta = __int_as_float(__builtin_amdgcn_readfirstlane(s_res[0]));
            mm_new = __builtin_amdgcn_readfirstlane(s_res[1]);
            if (__builtin_amdgcn_readfirstlane(s_res[2]) == WHY_NO_P) ++n_nop;
            if (r_delta < 0.0f) {
                acc_s = s;
                break;
            }
        }
        const int upto = acc_s >= 0 ? acc_s + 1 : qn;
        ++n_groups;
        n_steps += qn;
        n_useful += upto;
        n_nob += __popc(~flags & ((1u << upto) - 1u));
        if (acc_s < 0) {
            off += qn;
            continue;
        }
        // ---- the accept (Clustering.cpp:124-238, branch by branch) ----
        const int xx_acc = dr[acc_s], x_acc = cand_at_uniform(xx_acc), m_old = cand_at_uniform(mm_new);
        float d_new[PER], addend[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int pos = k + tid + 512 * u;
            d_new[u] = (pos < n && pos != xx_acc) ? a.D[sq_at(n, x_acc, y_pre[u])] : 0.0f;
            addend[u] = 0.0f;
        }
        float old_dn_xx = 0.0f;
        // the position that receives the replaced medoid: distances to the new medoid set, a fresh assignment -- wave 0
        if (wave == 0) {
            float dv[CLARANS_MAX_MEDOIDS / 64];
#pragma unroll
            for (int u = 0; u < CLARANS_MAX_MEDOIDS / 64; ++u) {
                const int mm = lane + 64 * u;
                dv[u] = FLT_MAX;
                if (mm < k) {
                    dv[u] = a.D[sq_at(n, mm == mm_new ? x_acc : cand_at(mm), m_old)];
                    a.DMt[(size_t)mm * n + xx_acc] = dv[u];
                }
            }
            float v1 = FLT_MAX, v2 = FLT_MAX;
            int i1 = INT_MAX, i2 = INT_MAX;
#pragma unroll
            for (int u = 0; u < CLARANS_MAX_MEDOIDS / 64; ++u) {
                const int mm = lane + 64 * u;
                if (mm < k && (dv[u] < v1 || i1 == INT_MAX)) { v1 = dv[u]; i1 = mm; }
            }
            wave_first_min_valid(v1, i1);
#pragma unroll
            for (int u = 0; u < CLARANS_MAX_MEDOIDS / 64; ++u) {
                const int mm = lane + 64 * u;
                if (mm < k && mm != i1 && (dv[u] < v2 || i2 == INT_MAX)) { v2 = dv[u]; i2 = mm; }
            }
            wave_first_min_valid(v2, i2);
            if (lane == 0) {
                const bool has1 = i1 != INT_MAX && v1 < FLT_MAX, has2 = i2 != INT_MAX && v2 < FLT_MAX;
                s_xx_state = pack_state(has1 ? v1 : FLT_MAX, has2 ? v2 : FLT_MAX, has1 ? i1 : -1, has2 ? i2 : -1);
            }
        }
        __syncthreads();
        bool need[PER]; // this position has to look at all k slots again
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int pos = k + tid + 512 * u;
            need[u] = false;
            if (pos >= n) continue;
            if (pos == xx_acc) {
                old_dn_xx = s_pre[u].x;
                s_pre[u] = s_xx_state;
                y_pre[u] = m_old;
                addend[u] = s_pre[u].x;
                continue;
            }
            apply_accept_to_position(s_pre[u], d_new[u], mm_new, addend[u], need[u]);
        }
        rescan_positions<PER>(a, need, d_new, s_pre, mm_new);
        // the new medoid's row of the member-to-medoid matrix, the accept's cost addends in position order
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int pos = k + tid + 512 * u;
            if (pos >= n) continue;
            a.cost_log[1 + pos - k] = addend[u];
            if (pos == xx_acc) a.cost_log[0] = -old_dn_xx;
            else a.DMt[(size_t)mm_new * n + pos] = d_new[u];
        }
        __syncthreads(); // (the order's old entries have been read; the log is where cost_accumulate reads it; the staging is free)
#pragma unroll
        for (int u = 0; u < PER; ++u)
            if (k + tid + 512 * u < n) s_dn[y_pre[u]] = s_pre[u].x;
        if (tid == 0) {
            __hip_atomic_store(a.cand + mm_new, x_acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(a.cand + xx_acc, m_old, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            s_dn[x_acc] = 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < MEMBERS / 512; ++j) dnm[j] = tid + 512 * j < n ? s_dn[tid + 512 * j] : -FLT_MAX;
        cost = cost_accumulate(a.cost_log, 1 + cnt, cost, reinterpret_cast<float*>(s_e), reinterpret_cast<float*>(s_we));
        P += off + acc_s + 1;
        off = 0;
        first = 0;
        ++accepts;
    }
    // ---- the state where the host (or the next launch) finds it ----
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int pos = k + tid + 512 * u;
        if (pos >= n) continue;
        a.cand[pos] = y_pre[u];
        a.st[pos] = s_pre[u];
    }
    __syncthreads(); // (the order and the states are where the next search of the chain, or the next launch, finds them)
    R.P = P;
    R.off = off;
    R.first = first;
    R.accepts = accepts;
    R.fresh = 0;
    R.cost = cost;
    R.n_groups = n_groups;
    R.n_steps = n_steps;
    R.n_useful = n_useful;
    R.n_nob = n_nob;
    R.n_nop = n_nop;
    return status;
}

// ---------------------------------------------------------------------------------------------------------------
// A CHAIN OF LOCAL SEARCHES IN ONE WORKGROUP (round 6): CLARANS::operator() from its first shuffle to its last comparison
// (Clustering.cpp:31-258) without the host in between.  Per local search: the candidate order is permuted (the
// partial_shuffle of cpp:46 moves POSITIONS, whatever they hold, and its generator does not look at the search: the host
// hands over the composed permutation), the search starts (chain_init), runs (clarans_search_body), and its medoids
// replace the kept ones if it is strictly cheaper (cpp:239-257).  All splits of a level of the FastTree recursion run as
// one launch, a workgroup each -- up to 512 resident at once -- and nothing else competes for the CUs meanwhile
// (lcsgpu_clarans_batch).  A workgroup leaves early only for want of pre-drawn positions or at the end of a time slice
// (a test aid); the state block says where it was.
template <int KPT>
__global__ __launch_bounds__(512, 4) void clarans_chain_kernel(const ClaransChain* __restrict__ chains, long long slice_ticks)
{
    const ClaransChain& ch = chains[blockIdx.x];
    const ClaransArgs& a = ch.a;
    const long long t_begin = wall_clock64();
    __shared__ int s_better;
    const int tid = threadIdx.x, k = a.n_medoids, n = a.n_elems;
    __builtin_amdgcn_s_setprio(3);
    SearchRegs R;
    R.P = __builtin_amdgcn_readfirstlane(a.state[ST_P]);
    R.off = __builtin_amdgcn_readfirstlane(a.state[ST_OFF]);
    R.first = __builtin_amdgcn_readfirstlane(a.state[ST_FIRST]);
    R.accepts = __builtin_amdgcn_readfirstlane(a.state[ST_ROUNDS]);
    R.fresh = __builtin_amdgcn_readfirstlane(a.state[ST_FRESH]);
    R.cost = __int_as_float(a.state[ST_COST]);
    R.n_groups = a.state[ST_N_ROUNDS];
    R.n_steps = a.state[ST_N_STEPS];
    R.n_useful = a.state[ST_N_USEFUL];
    R.n_nob = a.state[ST_N_NOB];
    R.n_nop = a.state[ST_N_NOP];
    int iter = __builtin_amdgcn_readfirstlane(a.state[ST_ITER]), need_init = __builtin_amdgcn_readfirstlane(a.state[ST_NEED_INIT]);
    float best = __int_as_float(a.state[ST_BEST]); // (kept by thread 0)
    int status = 0;
    while (iter < ch.num_local) {
        if (need_init) {
            // the order before this search: position i takes what position perm[i] held (the first search starts from 0, 1, 2, ...)
            const int32_t* perm = ch.perm + (size_t)iter * (size_t)n;
            int v[(CLARANS_MAX_MEDOIDS + CLARANS_MAX_NONMEDOIDS) / 512];
#pragma unroll
            for (int u = 0; u < (CLARANS_MAX_MEDOIDS + CLARANS_MAX_NONMEDOIDS) / 512; ++u) {
                const int i = tid + 512 * u;
                if (i < n) {
                    const int from = perm[i];
                    v[u] = iter == 0 ? from : __hip_atomic_load(a.cand + from, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < (CLARANS_MAX_MEDOIDS + CLARANS_MAX_NONMEDOIDS) / 512; ++u) {
                const int i = tid + 512 * u;
                if (i < n) __hip_atomic_store(a.cand + i, v[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            __syncthreads();
            chain_init(a);
            __syncthreads();
            R.off = 0;
            R.first = 1;
            R.fresh = 1;
            R.cost = 0.0f;
            need_init = 0;
        }
        status = clarans_search_body<KPT>(a, R, slice_ticks, t_begin);
        if (status != 1) break;
        // the search is over: its medoids stand if it is strictly cheaper than the best so far (Clustering.cpp:239-257)
        if (tid == 0) {
            s_better = R.cost < best;
            if (R.cost < best) best = R.cost;
        }
        __syncthreads();
        if (s_better)
            for (int mm = tid; mm < k; mm += 512) ch.best[mm] = __hip_atomic_load(a.cand + mm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();
        ++iter;
        need_init = 1;
    }
    if (tid == 0) {
        int32_t* st = a.state;
        st[ST_P] = R.P;
        st[ST_DONE] = iter >= ch.num_local;
        st[ST_ROUNDS] = R.accepts;
        st[ST_FRESH] = R.fresh;
        st[ST_COST] = __float_as_int(R.cost);
        st[ST_ERR] = 0;
        st[ST_MORE_DRAWS] = status == 2;
        st[ST_OFF] = R.off;
        st[ST_FIRST] = R.first;
        st[ST_N_ROUNDS] = R.n_groups;
        st[ST_N_STEPS] = R.n_steps;
        st[ST_N_USEFUL] = R.n_useful;
        st[ST_N_NOB] = R.n_nob;
        st[ST_N_NOP] = R.n_nop;
        st[ST_ITER] = iter;
        st[ST_BEST] = __float_as_int(best);
        st[ST_NEED_INIT] = need_init;
        st[ST_TICKS] += (int)(wall_clock64() - t_begin);
        st[ST_CU] = (int)((__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 8) | (__builtin_amdgcn_s_getreg((4 << 0) | (8 << 6) | (7 << 11))));
    }
}

// the samples' float distances as full symmetric matrices, D[i * n + j] = D[j * n + i] = transform(LCS(ref = ids[i], partner =
// ids[j])), j < i (Transform<float, ...>, tree/AbstractTreeGenerator.hpp:28-82, the float table built at upload); all samples in
// one launch (grid.y = sample)
template <typename T>
__global__ __launch_bounds__(256) void subset_dist_batch_kernel(const T* __restrict__ lcs, const ClaransChain* __restrict__ chains,
                                                                const uint32_t* __restrict__ lens, const float* __restrict__ pow_f32, int kind)
{
    const ClaransChain& ch = chains[blockIdx.y];
    const int n = ch.a.n_elems, i = blockIdx.x + 1;
    if (i >= n) return;
    float* D = const_cast<float*>(ch.a.D);
    const int32_t* ids = ch.ids;
    const T* tri = lcs + ch.tri0;
    if (threadIdx.x == 0) {
        D[(size_t)i * n + i] = 0.0f; // (never read)
        if (i == 1) D[0] = 0.0f;
    }
    const uint32_t len_i = lens[ids[i]];
    const size_t row = (size_t)i * (i - 1) / 2;
    for (int j = threadIdx.x; j < i; j += 256) {
        const uint32_t l = tri[row + j];
        const uint32_t indel = len_i + lens[ids[j]] - 2u * l;
        float d;
        if (l == 0) d = FLT_MAX;
        else if (kind == 1) d = __fdiv_rn(pow_f32[indel], (float)l);
        else d = __fdiv_rn((float)indel, (float)l);
        D[(size_t)i * n + j] = d;
        D[(size_t)j * n + i] = d;
    }
}

hipError_t launch_subset_distances_batch(const void* lcs, int elem_size, const ClaransChain* chains, int n_chains, int max_n,
                                         const uint32_t* lens, const float* pow_f32, int kind, hipStream_t stream)
{
    if (n_chains <= 0 || max_n < 2) return hipSuccess;
    const dim3 grid((unsigned)(max_n - 1), (unsigned)n_chains);
    if (elem_size == 2) hipLaunchKernelGGL(subset_dist_batch_kernel<uint16_t>, grid, dim3(256), 0, stream, (const uint16_t*)lcs, chains, lens, pow_f32, kind);
    else hipLaunchKernelGGL(subset_dist_batch_kernel<uint32_t>, grid, dim3(256), 0, stream, (const uint32_t*)lcs, chains, lens, pow_f32, kind);
    return hipGetLastError();
}

// every chain in its own workgroup, to its end (or for `slice_us` microseconds, 0 = no limit)
hipError_t launch_clarans_chains(const ClaransChain* chains, int n_chains, int max_medoids, int slice_us, hipStream_t stream)
{
    if (n_chains <= 0) return hipSuccess;
    const int kpt = std::max(1, ((max_medoids + 7) / 8 + 63) / 64);
    const long long ticks = slice_us > 0 ? (long long)slice_us * 100 : LLONG_MAX; // wall_clock64: 100 MHz
    if (kpt <= 1) hipLaunchKernelGGL(clarans_chain_kernel<1>, dim3(n_chains), dim3(512), 0, stream, chains, ticks);
    else hipLaunchKernelGGL(clarans_chain_kernel<2>, dim3(n_chains), dim3(512), 0, stream, chains, ticks);
    return hipGetLastError();
}

} // namespace lcsgpu
