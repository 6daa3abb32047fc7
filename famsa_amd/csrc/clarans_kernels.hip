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
//                         slot, every lane accumulates deltas[slot] over the non-medoids in ascending
//                         position -- the reference's float additions in the reference's order;
//   clarans_apply_kernel  takes the FIRST step of the window whose best delta is negative, swaps, and
//                         re-derives nearest / second-nearest medoid of every non-medoid exactly as
//                         the reference's update branch does (one lane per non-medoid).
// One round = these two launches -- or, since round 4 and where every position's state fits the registers of one
// workgroup, ONE: clarans_round_kernel applies the previous round's accept inside every step's workgroup and then
// evaluates (below).  The host enqueues rounds in batches and looks at the `done` flag between batches.  Ties, comparison directions and float operation order follow the reference
// line by line; the running cost is summed sequentially from a per-round log of its addends.
//
// Layout: D = float triangle over the sample members (D[i*(i-1)/2 + j], j < i).  All search state is
// kept per candidate POSITION (not per member), so the lanes of a wave read it coalesced:
// st[pos] = nearest / second-nearest bookkeeping, DMt[mm*n + pos] = distance of the member at pos to
// the medoid in slot mm (kept in step with the swaps so that the reference's updateAssignment scan
// is k coalesced loads instead of k scattered triangle entries per lane).
#include <hip/hip_runtime.h>

#include <cfloat>
#include <algorithm>
#include <climits>
#include <cstdint>

#include "lcs_kernels.h"

namespace lcsgpu {

namespace {

__device__ __forceinline__ size_t tri_at(int i, int j)
{
    return i >= j ? (size_t)j + (size_t)i * (i - 1) / 2 : (size_t)i + (size_t)j * (j - 1) / 2;
}

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

enum { ST_P = 0, ST_DONE = 1, ST_LOG_LEN = 2, ST_ROUNDS = 3, ST_ARRIVE = 4, ST_COST = 5, ST_ERR = 6, ST_WIN = 7,
       ST_OFF = 8, ST_STAGE = 9, ST_FIRST = 10,
       ST_N_ROUNDS = 11, ST_N_STEPS = 12, ST_N_USEFUL = 13,
       ST_N_COMMON = 14, ST_N_GENERAL = 15 }; // list evaluations: entries that add to every other slot (summed over the steps), steps that took the general walk // statistics of the search (LCSGPU_PROFILE): rounds, steps evaluated, steps up to the accepted one

// A window of pending steps is evaluated in stages of 16, 32, 64, 64, ... steps (ClaransArgs::stage0 = 16,
// LCSGPU_CLARANS_STAGE0): the steps after the accepted one are wasted work that other searches running at
// the same time have to queue behind.  Measured at 3 x 10^6 sequences (LCSGPU_PROFILE prints the counts): 878
// searches, 282 000 rounds, 224 000 accepts, 5.9 M steps evaluated of which 2.7 M up to the accepted one (the
// accepted step is the 8th of its round on average); a first stage of 4 / 8 / 16 / 24 steps gives a tree stage
// of 2.37 / 2.20 / 2.09-2.14 / 2.05 s -- the rate of dependent rounds, not the evaluation work, is the limit.
constexpr int STAGE_MAX = 64; // = the apply kernel's workgroup size: one result per lane
__device__ __forceinline__ int window_size(int corrected, int first) { return first ? corrected : (corrected > 0 ? corrected - 1 : 0); }
__device__ __forceinline__ int stage_size(int stage, int left, int stage0)
{
    const int want = min(STAGE_MAX, stage0 << min(stage, 6));
    return left < want ? (left < 0 ? 0 : left) : want;
}

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

// Start of one local search (Clustering.cpp:49-79): every non-medoid's distances to the medoid
// slots (DMt[mm * n + pos]) and assignment, the addends of the initial cost in position order, and
// the first window of pending steps (position and member of each).
__global__ __launch_bounds__(256) void clarans_init_kernel(ClaransArgs a)
{
    const int W = a.corrected;
    const int pos = blockIdx.x * 256 + threadIdx.x;
    const int k = a.n_medoids, n = a.n_elems;
    const int p = a.state[ST_P];
    if (pos == 0) {
        a.state[ST_DONE] = 0;
        a.state[ST_LOG_LEN] = n - k;
        a.state[ST_ROUNDS] = 0;
        a.state[ST_ARRIVE] = a.fused ? 1 : 0; // (fused rounds: "no round has run yet")
        a.state[ST_COST] = __float_as_int(0.0f);
        a.state[ST_WIN] = 0;
        a.state[ST_OFF] = 0;
        a.state[ST_STAGE] = 0;
        a.state[ST_FIRST] = 1;
        a.state[ST_N_ROUNDS] = 0;
        a.state[ST_N_STEPS] = 0;
        a.state[ST_N_USEFUL] = 0;
        a.state[ST_N_COMMON] = 0;
        a.state[ST_N_GENERAL] = 0;
        if (p + W > a.draws_len) a.state[ST_ERR] = 1;
    }
    if (p + W <= a.draws_len)
        for (int j = pos; j < W; j += gridDim.x * 256) {
            const int xx = a.draws[p + j];
            a.win_xx[j] = xx;
            a.win_x[j] = a.cand[xx];
        }
    if (pos < k || pos >= n) return;
    const int y = a.cand[pos];
    Nearest2 nb;
#pragma unroll 4
    for (int mm = 0; mm < k; ++mm) {
        const float d = a.D[tri_at(a.cand[mm], y)];
        a.DMt[(size_t)mm * n + pos] = d;
        nb.feed(d, mm);
    }
    a.st[pos] = pack_state(nb.dn, nb.ds, nb.an, nb.as);
    a.cost_log[pos - k] = nb.dn;
}

// loads of what another workgroup of the SAME launch wrote (the one-XCD kernel below): they bypass this CU's L1 (sc1)
template <typename T>
__device__ __forceinline__ T ldc(const T* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float4 ldc4(const float4* p)
{
    const uint64_t* q = reinterpret_cast<const uint64_t*>(p);
    const uint64_t lo = ldc(q), hi = ldc(q + 1);
    return make_float4(__uint_as_float((uint32_t)lo), __uint_as_float((uint32_t)(lo >> 32)), __uint_as_float((uint32_t)hi),
                       __uint_as_float((uint32_t)(hi >> 32)));
}
template <bool COH, typename T>
__device__ __forceinline__ T ld(const T* p) { return COH ? ldc(p) : *p; }
template <bool COH>
__device__ __forceinline__ float4 ld4(const float4* p) { return COH ? ldc4(p) : *p; }

// running cost: c += addend for every logged addend, in order; zeros are the identity (c starts at +0.0f and can
// never become -0.0f), so only the others are walked.  Valid in thread 0.
template <bool COH>
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
            v[u] = t < cnt ? ld<COH>(&cost_log[c0 + t]) : 0.0f;
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
// (cpp:121-122); the result is valid in thread 0.  Every slot's delta is a sequential float sum over the non-medoids in
// position order.  A non-medoid adds to its own nearest slot always and to all others only when the candidate is closer
// to it than its medoid (rare), so each of the 8 waves -- wave w owns the slots [w * kpw, (w + 1) * kpw) -- first
// compacts, in order, the entries that can change one of ITS slots and then walks only those: skipped entries would add
// +0.0f, the identity.  y_pre / s_pre: the first chunk's members and states when the caller has requested them already
// (PRE), else loaded here.  TIMED: phase timers (loads, staging, own walk, slowest wave, reduction; ticks of 10 ns).
template <int KPT, bool COH, bool PRE, bool TIMED>
__device__ __forceinline__ void evaluate_step(const ClaransArgs& a, int xx, int x, int* y_pre, float4* s_pre, float4* s_e,
                                              float4 (*s_we)[128], float& best_out, int& bk_out, unsigned long long* t_ev)
{
    constexpr int CH = 2048, PER = CH / 512, HALF = CH / 2, SUB = 128;
    unsigned long long te0 = TIMED ? wall_clock64() : 0;
    auto elap = [&](int ph) {
        if (TIMED) {
            const unsigned long long t1 = wall_clock64();
            t_ev[ph] += t1 - te0;
            te0 = t1;
        }
    };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = a.n_medoids, n = a.n_elems;
    const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    const int kpw = (k + 7) >> 3;
    const int klo = wave * kpw, khi = min(k, klo + kpw);
    float acc[KPT];
    int slot[KPT];
#pragma unroll
    for (int q = 0; q < KPT; ++q) {
        acc[q] = 0.0f;
        slot[q] = klo + lane + 64 * q;
    }
    for (int c0 = k; c0 < n; c0 += CH) {
        const int cnt = min(CH, n - c0);
        // entries of this chunk: (addend for the own slot, addend for the other slots, own slot)
        float dxy[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) { // the gathers from the triangle, all in flight together
            const int t = tid + 512 * u;
            if (!PRE || c0 != k) {
                y_pre[u] = t < cnt ? ld<COH>(&a.cand[c0 + t]) : 0;
                s_pre[u] = t < cnt ? ld4<COH>(&a.st[c0 + t]) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            dxy[u] = (t < cnt && c0 + t != xx) ? a.D[tri_at(x, y_pre[u])] : 0.0f;
        }
        float4 ent[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int t = tid + 512 * u;
            ent[u] = make_float4(0.0f, 0.0f, __int_as_float(-1), 0.0f); // position xx: contributes nothing
            if (t < cnt && c0 + t != xx) {
                const float dn = s_pre[u].x, ds = s_pre[u].y;
                const float m = ds < dxy[u] ? ds : dxy[u]; // std::min(dxy, ds)
                const float change = __fsub_rn(dxy[u], dn);
                ent[u].x = __fsub_rn(m, dn);                     // goes to deltas[nearest(y)]
                ent[u].y = change < 0.0f ? change : 0.0f;        // goes to every other slot when negative
                ent[u].z = s_pre[u].z;
            }
        }
        if (TIMED && ent[0].x == 12345.678f) __builtin_amdgcn_s_sleep(1); // (the loads have arrived)
        elap(0);
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
            elap(1);
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
            elap(2); // this wave's own walk
            __syncthreads();
            elap(3); // waiting for the slowest wave
        }
    }
    // std::min_element over slots [n_fixed, k): smallest value, earliest slot among equals
    float* s_v = reinterpret_cast<float*>(&s_we[0][0]);
    int* s_k = reinterpret_cast<int*>(&s_we[1][0]);
    float best = 0.0f;
    int bk = INT_MAX;
#pragma unroll
    for (int q = 0; q < KPT; ++q)
        if (slot[q] >= a.n_fixed && slot[q] < khi && (bk == INT_MAX || acc[q] < best)) { best = acc[q]; bk = slot[q]; }
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
    }
    elap(4);
}

// ---- the same evaluation with PER-SLOT LISTS (round 4) ------------------------------------------------------------------
// evaluate_step's walk hands every entry to all 64 lanes of the wave that owns its slot -- a broadcast read of 16 B by 64
// lanes per entry and wave, 1 KB through the CU's LDS return port: 10.6 of an evaluation's 18 us, and what two or three
// evaluations resident on one CU fight over.  But a slot's delta only ever adds (a) the entries of ITS
// members, and (b) the few entries whose member is closer to the candidate than to its own medoid (they add to every
// other slot).  So: sort the entries by slot, stably (= by position inside a slot), once per evaluation, and let lane m
// walk slot m's own run -- every lane reads a DIFFERENT 8 bytes per step -- merging in the short list (b) by position.
// The additions each slot sees are the same, in the same order; only who reads what has changed.
//   1. entries in position order, 4 per thread (as before); list (b) is compacted in position order (ballots);
//   2. per 64-position chunk (one wave, four chunks each) every lane finds its rank among the lanes holding the same
//      slot -- a ballot per slot bit, nothing moves -- and the last lane of each slot leaves the count in LDS;
//   3. per slot: prefix of the counts over the 32 chunks (chunk order = position order), prefix of the totals over the
//      slots = where each slot's run starts; every lane scatters its (addend, position) to its place;
//   4. lane l < 32 of wave w walks slot 8 l + w (k <= 256).
// Shapes outside (more than 2048 non-medoids, more than 256 medoids, more than 256 entries in list (b)) take
// evaluate_step.  Returns false when the caller has to do that.
// MEASURED (round 4, 2000 members / 100 medoids, profiles/clarans_lists_r04.txt): bit-identical on all 23 shapes and the
// 3 x 10^6-sequence tree, an evaluation's phases sum to 15 us on average (loads 1.4, ranks 1.7, prefixes + scatter 1.4,
// walks 5.1, waiting for the slowest wave 5.6 -- the largest cluster's chain) against 17.7 us for the broadcast walk --
// and the KERNEL takes 33.7 us against 19.9 us: a launch lasts as long as its slowest step, and a step whose candidate is
// closer to many members than their medoids pays per entry of list (b) in every lane, where the broadcast walk pays one
// more entry per wave.  So this stays opt-in (LCSGPU_CLARANS_LISTS=1); the default is evaluate_step.
constexpr int LISTS_MAX_S = 2048, LISTS_MAX_K = 256, LISTS_MAX_COMMON = 256;
// measurement aid (LCSGPU_CLARANS_LISTS=2): 10 ns ticks per phase of evaluate_step_lists, summed over the evaluations
// of workgroup 0's thread 0, and their count at [7]
__device__ unsigned long long g_lists_ticks[8];
template <bool TIMED>
__device__ __forceinline__ bool evaluate_step_lists(const ClaransArgs& a, int xx, int x, const int* y_pre, const float4* s_pre,
                                                    float4* s_e, float4 (*s_we)[128], uint32_t* s_misc, float& best_out, int& bk_out,
                                                    unsigned long long* t_ev)
{
    unsigned long long te0 = TIMED ? wall_clock64() : 0;
    auto elap = [&](int ph) {
        if (TIMED) {
            const unsigned long long t1 = wall_clock64();
            t_ev[ph] += t1 - te0;
            te0 = t1;
        }
    };
    constexpr int PER = 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = a.n_medoids, n = a.n_elems;
    const int S = n - k;
    const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    // LDS: s_e (16 KB) = the sorted (addend, position) pairs; s_we (16 KB) = counts per (chunk, slot) as uint16;
    // s_misc (4 KB) = run starts [k + 1], list (b): positions, slots, values [256 each], its per-chunk counts [32]
    float2* s_ent = reinterpret_cast<float2*>(s_e);
    uint16_t* s_cnt = reinterpret_cast<uint16_t*>(&s_we[0][0]); // [32][k]
    uint32_t* s_start = s_misc;                                     // [257]
    uint16_t* s_cpos = reinterpret_cast<uint16_t*>(s_misc + 260);   // [256]
    uint16_t* s_cslot = s_cpos + LISTS_MAX_COMMON;                  // [256]
    float* s_cval = reinterpret_cast<float*>(s_misc + 260 + 256);   // [256]
    uint32_t* s_cc = s_misc + 260 + 512;                            // [32] + total at [32]
    // ---- 1. the entries, in position order ----
    float dxy[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int t = tid + 512 * u;
        dxy[u] = (t < S && k + t != xx) ? a.D[tri_at(x, y_pre[u])] : 0.0f;
    }
    for (int i = tid; i < 32 * k / 2; i += 512) reinterpret_cast<uint32_t*>(s_cnt)[i] = 0u; // (k even or not: the tail word below)
    if (tid == 0 && (k & 1)) s_cnt[32 * k - 1] = 0;
    float own[PER], other[PER];
    int slot[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int t = tid + 512 * u;
        own[u] = 0.0f;
        other[u] = 0.0f;
        slot[u] = -1; // position xx and the places behind the end: no entry
        if (t < S && k + t != xx) {
            const float dn = s_pre[u].x, ds = s_pre[u].y;
            const float m = ds < dxy[u] ? ds : dxy[u]; // std::min(dxy, ds)
            const float change = __fsub_rn(dxy[u], dn);
            own[u] = __fsub_rn(m, dn);                  // goes to deltas[nearest(y)]
            other[u] = change < 0.0f ? change : 0.0f;   // goes to every other slot when negative
            slot[u] = __float_as_int(s_pre[u].z);
        }
    }
    elap(0);
    __syncthreads(); // the counts are zero
    // ---- 2. per chunk (chunk u * 8 + wave = positions 512 u + 64 wave ...): list (b) counts; my rank among the lanes of
    // the chunk that hold the same slot, in lane = position order.  "The lanes with my slot" = the AND over the slot's bits
    // of (lanes whose bit is set) or its complement: a ballot and two selects per bit, nothing moves between lanes. ----
    int nbits = 1;
    while ((1 << nbits) < k) ++nbits;
    int rank[PER];
    uint64_t cmask[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        cmask[u] = __ballot(other[u] < 0.0f);
        if (lane == 0) s_cc[u * 8 + wave] = (uint32_t)__popcll(cmask[u]);
        const bool ent = slot[u] >= 0;
        uint64_t same = __ballot(ent);
        for (int bit = 0; bit < nbits; ++bit) {
            const bool one = ((slot[u] >> bit) & 1) != 0;
            const uint64_t ones = __ballot(one);
            same &= one ? ones : ~ones;
        }
        rank[u] = __popcll(same & lt_mask);
        const bool is_last = ent && (lane == 63 || (same >> (lane + 1)) == 0ull);
        if (is_last) s_cnt[(u * 8 + wave) * k + slot[u]] = (uint16_t)__popcll(same);
    }
    __syncthreads();
    elap(1);
    // ---- 3. where every (chunk, slot) piece goes ----
    if (tid < k) { // counts -> exclusive prefix over the chunks, in place; the slot's total (all reads first: one LDS round trip)
        uint32_t h[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) h[c] = s_cnt[c * k + tid];
        uint32_t run = 0;
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            s_cnt[c * k + tid] = (uint16_t)run;
            run += h[c];
        }
        s_start[tid + 1] = run; // (totals for now)
    }
    if (wave == 7) { // list (b): exclusive prefix of the per-chunk counts (one wave, lane = chunk)
        const uint32_t h = lane < 32 ? s_cc[lane] : 0u;
        uint32_t incl = h;
#pragma unroll
        for (int d = 1; d < 32; d <<= 1) {
            const uint32_t o = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane - d) & 63) << 2, (int)incl);
            if (lane >= d) incl += o;
        }
        if (lane < 32) s_cc[lane] = incl - h;
        if (lane == 31) s_cc[32] = incl;
    }
    __syncthreads();
    if (wave == 0) { // inclusive scan of the totals over the slots: 4 per lane, then across the lanes
        uint32_t v[4], sum = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = lane * 4 + q;
            v[q] = m < k ? s_start[m + 1] : 0u;
            sum += v[q];
        }
        uint32_t incl = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = (uint32_t)__builtin_amdgcn_ds_bpermute(((lane - d) & 63) << 2, (int)incl);
            if (lane >= d) incl += o;
        }
        uint32_t run = incl - sum;
        if (lane == 0) s_start[0] = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int m = lane * 4 + q;
            run += v[q];
            if (m < k) s_start[m + 1] = run;
        }
    }
    __syncthreads();
    const int n_common = (int)s_cc[32];
    if (TIMED && tid == 0) { // (statistics; an atomic per evaluation on the search's state block costs microseconds per round)
        atomicAdd(&a.state[ST_N_COMMON], n_common);
        if (n_common > LISTS_MAX_COMMON) atomicAdd(&a.state[ST_N_GENERAL], 1);
    }
    if (n_common > LISTS_MAX_COMMON) return false; // (uniform) a candidate this central: the general walk
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int c = u * 8 + wave;
        if (slot[u] >= 0) {
            const uint32_t at = s_start[slot[u]] + s_cnt[c * k + slot[u]] + (uint32_t)rank[u];
            s_ent[at] = make_float2(own[u], __int_as_float(tid + 512 * u));
        }
        if (other[u] < 0.0f) { // list (b), in position order
            const uint32_t at = s_cc[c] + (uint32_t)__popcll(cmask[u] & lt_mask);
            s_cpos[at] = (uint16_t)(tid + 512 * u);
            s_cslot[at] = (uint16_t)slot[u];
            s_cval[at] = other[u];
        }
    }
    __syncthreads();
    elap(2);
    // ---- 4. lane l < 32 of wave w: the delta of slot 8 l + w (the slots of a wave are spread over the clusters) ----
    const int m = lane * 8 + wave;
    const bool has = lane < 32 && m < k;
    int i = has ? (int)s_start[m] : 0;
    const int e = has ? (int)s_start[m + 1] : 0;
    float acc = 0.0f;
    // A slot's additions are one dependent chain, and the largest cluster's chain is what the workgroup waits for: the
    // members are read eight at a time (independent ds_read_b64), and a batch that lies wholly in front of the next entry
    // of list (b) -- nearly all of them: that list holds a handful of entries -- is eight bare additions.
    // The window f[0..7] holds the members base .. base + 7 of my run (clamped to its end), the batch behind it is
    // prefetched while it is added; `off` of its members are consumed already.  An entry of list (b) costs a lane one
    // comparison unless members of its own lie in front of it (then: the window's additions, predicated).
    const int e_last = max(e - 1, 0);
    int base = i, off = 0;
    float2 f[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) f[q] = s_ent[min(base + q, e_last)];
    int next_pos = base < e ? __float_as_int(f[0].y) : 0x7FFFFFFF; // position of my next member
    for (int ci = 0; ci <= n_common; ++ci) {
        int bound = 0x7FFFFFFF, cs = -1;
        float cv = 0.0f;
        if (ci < n_common) { // (uniform: broadcast reads)
            bound = (int)s_cpos[ci];
            cs = (int)s_cslot[ci];
            cv = s_cval[ci];
        }
        while (next_pos < bound) { // members of my own in front of that entry
            if (off == 0 && base + 8 <= e && __float_as_int(f[7].y) < bound) { // the whole window: eight bare additions
                float2 g[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) g[q] = s_ent[min(base + 8 + q, e_last)];
#pragma unroll
                for (int q = 0; q < 8; ++q) acc = __fadd_rn(acc, f[q].x);
                base += 8;
#pragma unroll
                for (int q = 0; q < 8; ++q) f[q] = g[q];
                next_pos = base < e ? __float_as_int(f[0].y) : 0x7FFFFFFF;
            } else {
                bool go = true;
                int np = 0x7FFFFFFF;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const bool live = q >= off && base + q < e;                 // not consumed yet, inside my run
                    const bool take = go && live && __float_as_int(f[q].y) < bound;
                    acc = __fadd_rn(acc, take ? f[q].x : 0.0f); // (+0.0f is the identity: acc is never -0.0f)
                    off += take ? 1 : 0;
                    if (live && !take && go) np = __float_as_int(f[q].y); // the first member that stays
                    go = go && (take || !live);
                }
                if (off == 8 || base + off >= e) { // the window is used up
                    base += 8;
                    off = 0;
#pragma unroll
                    for (int q = 0; q < 8; ++q) f[q] = s_ent[min(base + q, e_last)];
                    np = base < e ? __float_as_int(f[0].y) : 0x7FFFFFFF;
                    if (base >= e) base = e;
                }
                next_pos = np;
            }
        }
        if (ci < n_common && has && cs != m) acc = __fadd_rn(acc, cv);
        // (an entry of list (b) that is my own member is added with its own-slot addend: positions are unique and
        //  `< bound` stops in front of it, so the next turn of the loop takes it)
    }
    elap(3);
    __syncthreads(); // every wave is done with the lists: the reduction below reuses s_we
    // std::min_element over slots [n_fixed, k): smallest value, earliest slot among equals
    float* s_v = reinterpret_cast<float*>(&s_we[0][0]);
    int* s_k = reinterpret_cast<int*>(&s_we[1][0]);
    float best = 0.0f;
    int bk = INT_MAX;
    if (has && m >= a.n_fixed) { best = acc; bk = m; }
    wave_first_min_valid(best, bk);
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
    }
    elap(4);
    return true;
}

// A kernel boundary leaves nothing in the caches that another XCD wrote, so every DEPENDENT global
// load of these small kernels costs a trip to memory (~1.5 us): both kernels are laid out to have as
// few dependent levels as possible -- everything whose address does not depend on the step is
// requested first, the pending steps' positions/members are precomputed by the previous kernel.

// One workgroup per pending step b of the window: evaluate_step for the member win_x[b].
// Workgroup W adds the previous round's cost addends to the running cost, in order.
template <int KPT>
__global__ __launch_bounds__(512) void clarans_eval_kernel(ClaransBatch batch)
{
    const ClaransArgs& a = batch.s[blockIdx.y];
    const int corrected = a.corrected;
    // 32 KB of LDS and 512 lanes per workgroup: four fit a CU, so the 65 x 16 workgroups of a full
    // batch of searches are resident together
    constexpr int CH = 2048, PER = CH / 512; // positions whose data a workgroup keeps in registers at a time
    __shared__ float4 s_e[CH / 2];      // 16 KB: the chunk's entries, staged in two halves
    __shared__ float4 s_we[8][128];     // 16 KB: per wave, the entries of one sub-chunk that concern its slots
    __shared__ uint32_t s_misc[1024];   //  4 KB: evaluate_step_lists' run starts and its short list
    const int b = blockIdx.x, tid = threadIdx.x;
    const int k = a.n_medoids, n = a.n_elems;
    // level 1: state, this step, and the first chunk's per-position data
    const int4 st0 = *reinterpret_cast<const int4*>(a.state);
    const int4 st1 = *reinterpret_cast<const int4*>(a.state + 4);
    const int4 st2 = *reinterpret_cast<const int4*>(a.state + 8);
    const int win = st1.w & 1;
    const bool cost_wg = b == (int)gridDim.x - 1;
    const int S = stage_size(st2.y, window_size(corrected, st2.z) - st2.x, a.stage0); // steps evaluated in this round
    if (!cost_wg && b >= S) return;
    const int bb = cost_wg ? 0 : st2.x + b; // index of this workgroup's step in the window
    const int xx = a.win_xx[win * a.win_cap + bb];
    const int x = a.win_x[win * a.win_cap + bb];
    int y_pre[PER];
    float4 s_pre[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int yy = k + tid + 512 * u;
        y_pre[u] = yy < n ? a.cand[yy] : 0;
        s_pre[u] = yy < n ? a.st[yy] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (st0.y) return; // done
    if (cost_wg) { // adds the previous round's cost addends to the running cost, in order
        const int len = st0.z;
        if (len == 0) return;
        const float c = cost_accumulate<false>(a.cost_log, len, __int_as_float(st1.y), reinterpret_cast<float*>(s_e), reinterpret_cast<float*>(s_we));
        if (tid == 0) a.state[ST_COST] = __float_as_int(c);
        return;
    }
    if (st1.z) return; // error flagged: apply ends the search
    float best = 0.0f;
    int bk = INT_MAX;
    bool evaluated = false;
    if (a.lists == 2 && n - k <= LISTS_MAX_S && k <= LISTS_MAX_K) {
        unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        evaluated = evaluate_step_lists<true>(a, xx, x, y_pre, s_pre, s_e, s_we, s_misc, best, bk, tk);
        if (tid == 0) { // (every step's workgroup: the kernel lasts as long as its slowest)
            for (int i = 0; i < 5; ++i) atomicAdd(&g_lists_ticks[i], tk[i]);
            atomicAdd(&g_lists_ticks[7], 1ull);
            if (!evaluated) atomicAdd(&g_lists_ticks[6], 1ull);
            unsigned long long tot = 0;
            for (int i = 0; i < 5; ++i) tot += tk[i];
            atomicMax(&g_lists_ticks[5], tot);
        }
    } else if (a.lists && n - k <= LISTS_MAX_S && k <= LISTS_MAX_K)
        evaluated = evaluate_step_lists<false>(a, xx, x, y_pre, s_pre, s_e, s_we, s_misc, best, bk, nullptr);
    if (!evaluated) {
        __syncthreads();
        evaluate_step<KPT, false, true, false>(a, xx, x, y_pre, s_pre, s_e, s_we, best, bk, nullptr);
    }
    if (tid == 0) {
        a.res_delta[b] = best;
        a.res_mm[b] = bk;
    }
}

// Accept the first improving step of the window (Clustering.cpp:124-238) or finish the search.
// Workgroups 0 .. gridDim-2 (one wave each): one lane per non-medoid position; the last workgroup
// rebuilds the position that receives the replaced medoid and prepares the next window of pending
// steps.  The workgroup that arrives last commits the swap.
constexpr int APPLY_MT = 128; // medoid slots staged per pass
__global__ __launch_bounds__(64) void clarans_apply_kernel(ClaransBatch batch)
{
    const ClaransArgs& a = batch.s[blockIdx.y];
    const int corrected = a.corrected;
    __shared__ int s_w;
    __shared__ int s_last;
    __shared__ float s_tile[APPLY_MT][64]; // 32 KB: [slot][lane]; the last workgroup uses it as one row
    int* st = a.state;
    const int tid = threadIdx.x;
    const int k = a.n_medoids, n = a.n_elems;
    const int n_wg = (n - k + 63) / 64 + 1; // this search's workgroups; the grid is sized for the largest search
    if ((int)blockIdx.x >= n_wg) return;
    const bool last_wg = (int)blockIdx.x == n_wg - 1;
    // level 1: everything whose address does not depend on the accepted step
    const int4 st0 = *reinterpret_cast<const int4*>(st);
    const int4 st1 = *reinterpret_cast<const int4*>(st + 4);
    const int4 st2 = *reinterpret_cast<const int4*>(st + 8);
    const int W = window_size(corrected, st2.z), off = st2.x;
    const int S = stage_size(st2.y, W - off, a.stage0);
    const int W_next = window_size(corrected, 0);
    const int yy = k + blockIdx.x * 64 + tid;
    const bool have = !last_wg && yy < n;
    const int y_mine = have ? a.cand[yy] : 0;
    int med_pre[CLARANS_MAX_MEDOIDS / 64]; // last workgroup: the current medoids, slots tid, tid + 64, ...
#pragma unroll
    for (int u = 0; u < CLARANS_MAX_MEDOIDS / 64; ++u) med_pre[u] = (last_wg && tid + 64 * u < k) ? a.cand[tid + 64 * u] : 0;
    const float4 s = have ? a.st[yy] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (have) {
        const float* col = a.DMt + yy;
        const int k0 = min(k, APPLY_MT);
#pragma unroll 16
        for (int mm = 0; mm < k0; ++mm) s_tile[mm][tid] = col[(size_t)mm * n];
    }
    const float rd_mine = a.res_delta[tid]; // a stage has at most 64 steps: one per lane (the array holds >= 64)
    if (st0.y) return; // done
    if (tid == 0) s_w = INT_MAX;
    __syncthreads();
    if (!st1.z && tid < S && rd_mine < 0.0f) atomicMin(&s_w, tid);
    __syncthreads();
    const int w = s_w;
    const int p = st0.x;
    const int win = st1.w & 1;
    const bool accept = w != INT_MAX;
    const int j = off + (accept ? w : 0); // index of the accepted step in the window
    // level 2
    const int xx = a.win_xx[win * a.win_cap + j];
    const int x = a.win_x[win * a.win_cap + j]; // the new medoid
    const int mm_new = accept ? a.res_mm[w] : 0;
    int m_old = 0;
    if (!accept) {
        // no accept among this round's steps: the next round takes the next stage of the window
    } else if (last_wg) {
        // level 3: the medoid that is replaced; from now on it sits at position xx
        m_old = a.cand[mm_new];
        // next window of pending steps, against the candidate order after this swap
        const int p_new = p + j + 1;
        if (p_new + W_next > a.draws_len) {
            if (tid == 0) st[ST_ERR] = 1;
        } else {
            int32_t* nxx = a.win_xx + (1 - win) * a.win_cap;
            int32_t* nx = a.win_x + (1 - win) * a.win_cap;
            for (int j = tid; j < W_next; j += 64) {
                const int xn = a.draws[p_new + j];
                nxx[j] = xn;
                nx[j] = xn == xx ? m_old : a.cand[xn];
            }
        }
        // position xx: cost -= dists_nearest[new medoid] first (cpp:131), then the replaced medoid
        // gets its distances to the new medoid set and a fresh assignment (cpp:150-157)
        const float old_dn = a.st[xx].x;
        // updateAssignment over the new medoid set = first minimum over the slots, then first minimum
        // over the slots without that one (what the sequential scan of cpp:262-305 arrives at)
        float dv[CLARANS_MAX_MEDOIDS / 64];
#pragma unroll
        for (int u = 0; u < CLARANS_MAX_MEDOIDS / 64; ++u) {
            const int mm = tid + 64 * u;
            dv[u] = FLT_MAX;
            if (mm < k) {
                dv[u] = a.D[tri_at(mm == mm_new ? x : med_pre[u], m_old)];
                a.DMt[(size_t)mm * n + xx] = dv[u];
            }
        }
        float v1 = FLT_MAX, v2 = FLT_MAX;
        int i1 = INT_MAX, i2 = INT_MAX;
#pragma unroll
        for (int u = 0; u < CLARANS_MAX_MEDOIDS / 64; ++u) {
            const int mm = tid + 64 * u;
            if (mm < k && (dv[u] < v1 || i1 == INT_MAX)) { v1 = dv[u]; i1 = mm; }
        }
        wave_first_min_valid(v1, i1);
#pragma unroll
        for (int u = 0; u < CLARANS_MAX_MEDOIDS / 64; ++u) {
            const int mm = tid + 64 * u;
            if (mm < k && mm != i1 && (dv[u] < v2 || i2 == INT_MAX)) { v2 = dv[u]; i2 = mm; }
        }
        wave_first_min_valid(v2, i2);
        if (tid == 0) {
            // the scan starts from (FLT_MAX, -1): a slot at FLT_MAX never replaces it
            const bool has1 = i1 != INT_MAX && v1 < FLT_MAX, has2 = i2 != INT_MAX && v2 < FLT_MAX;
            a.st[xx] = pack_state(has1 ? v1 : FLT_MAX, has2 ? v2 : FLT_MAX, has1 ? i1 : -1, has2 ? i2 : -1);
            a.cost_log[0] = -old_dn;
            a.cost_log[1 + xx - k] = has1 ? v1 : FLT_MAX;
        }
    } else if (accept && have && yy != xx) {
        const float d_new = a.D[tri_at(x, y_mine)]; // level 3
        a.DMt[(size_t)mm_new * n + yy] = d_new;
        const float dn_y = s.x, ds_y = s.y;
        const int an_y = __float_as_int(s.z), as_y = __float_as_int(s.w);
        float addend = 0.0f;
        bool rescan = false;
        float4 out = s;
        if (an_y == mm_new) { // its medoid is the one that left
            if (d_new < ds_y) {
                out.x = d_new;
                addend = __fsub_rn(d_new, dn_y);
            } else {
                rescan = true;
                addend = __fsub_rn(ds_y, dn_y);
            }
        } else if (d_new < dn_y) {
            out = pack_state(d_new, dn_y, mm_new, an_y);
            addend = __fsub_rn(d_new, dn_y);
        } else if (as_y != mm_new && d_new < ds_y) {
            out.y = d_new;
            out.w = __int_as_float(mm_new);
        } else if (as_y != mm_new && d_new > ds_y) {
            // Neither of its two nearest slots is the one that changed, and the new medoid is strictly farther than
            // the second: updateAssignment (first minimum over the slots, then first minimum over the rest) gives what
            // it gave before -- the reference rescans here (Clustering.cpp:228-232) and arrives at the same four
            // values.  (Equality with the second stays with the rescan: the slot order decides a tie.)
        } else {
            rescan = true;
        }
        if (rescan) {
            Nearest2 nb;
            const int k0 = min(k, APPLY_MT);
            for (int m0 = 0; m0 < k0; m0 += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = s_tile[min(m0 + u, k0 - 1)][tid];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (m0 + u < k0) nb.feed(m0 + u == mm_new ? d_new : v[u], m0 + u);
            }
            const float* col = a.DMt + yy;
            for (int m0 = APPLY_MT; m0 < k; m0 += APPLY_MT) { // more slots than one staging pass holds
                const int k1 = min(k, m0 + APPLY_MT);
#pragma unroll 16
                for (int mm = m0; mm < k1; ++mm) s_tile[mm - m0][tid] = col[(size_t)mm * n];
                for (int mm = m0; mm < k1; ++mm) nb.feed(mm == mm_new ? d_new : s_tile[mm - m0][tid], mm);
            }
            out = pack_state(nb.dn, nb.ds, nb.an, nb.as);
        }
        a.st[yy] = out;
        a.cost_log[1 + yy - k] = addend;
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(&st[ST_ARRIVE], 1) == n_wg - 1;
    __syncthreads();
    if (s_last && tid == 0) {
        st[ST_ARRIVE] = 0;
        st[ST_N_ROUNDS] += 1;
        st[ST_N_STEPS] += st1.z ? 0 : S;
        st[ST_N_USEFUL] += accept ? w + 1 : (st1.z ? 0 : S);
        if (accept) {
            const int mo = a.cand[mm_new];
            a.cand[mm_new] = x;
            a.cand[xx] = mo;
            st[ST_P] = p + j + 1;
            st[ST_LOG_LEN] = 1 + n - k;
            st[ST_ROUNDS] = st0.w + 1;
            st[ST_WIN] = 1 - win;
            st[ST_OFF] = 0;
            st[ST_STAGE] = 0;
            st[ST_FIRST] = 0;
        } else {
            st[ST_LOG_LEN] = 0;
            if (st1.z || off + S >= W) { // error, or `corrected` steps without an accept: this local search is over
                st[ST_P] = p + (st1.z ? 0 : W);
                st[ST_DONE] = 1;
            } else {
                st[ST_OFF] = off + S;
                st[ST_STAGE] = st2.y + 1;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// A ROUND AS ONE LAUNCH (round 4; LCSGPU_CLARANS_FUSED, the default where the shape allows).
//
// evaluate + apply are two dependent launches per round, and under load the apply -- 31 one-wave workgroups whose 3-4
// dependent load levels each wait behind the LCS launches and the other searches' evaluations -- costs 38 of a round's
// 101 us.  But what the apply does to a position is small (one gather, a handful of comparisons, rarely a rescan over the k
// slots), and the evaluation workgroups already hold every position's state in registers, four positions per thread.  So
// every step's workgroup APPLIES THE PREVIOUS ROUND'S ACCEPT ITSELF, to its own register copy of the state, and then
// evaluates its step of this round against it: one launch per round.
//   * All buffers a workgroup reads at its first load level and another writes in the same launch exist twice, by round
//     parity: state block, candidate order, per-position state, cost log, step results.  The parity-0 copies are the
//     arrays the two-launch form uses (what the host reads after an even number of rounds).
//   * Workgroup 0 is also the COMMITTER: it writes the applied state to the other parity (whole arrays: every thread its
//     four positions), the new column / row of the member-to-medoid matrix (in place: nobody consumes those entries in
//     the same launch -- a rescan overrides the changed slot, the replaced medoid's position is rebuilt from D), the
//     cost addends and the state block.  The last workgroup keeps the running cost, as before.
//   * The control flow of the two kernels (first improving step of the stage; stages 16, 32, 64, 64 ... of a window;
//     corrected / corrected - 1 steps without an accept end the search) is recomputed by every workgroup from the same
//     state block and step results; the arithmetic of the apply and of the evaluation is the two kernels', line by line.
// Shapes: n - k <= 2048 (the state of all positions in the registers of one workgroup), n > k.
enum { ST_FRESH = ST_ARRIVE }; // (the arrival counter of the two-launch form is free here) 1 = no round has run yet
template <int KPT>
__global__ __launch_bounds__(512) void clarans_round_kernel(ClaransBatch batch, int par, int last)
{
    const ClaransArgs& a = batch.s[blockIdx.y];
    constexpr int PER = 4;
    __shared__ float4 s_e[1024];        // 16 KB   evaluate_step's staging
    __shared__ float4 s_we[8][128];     // 16 KB
    __shared__ float4 s_xx_state;       // the rebuilt state of the position that received the replaced medoid
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = a.n_medoids, n = a.n_elems, corrected = a.corrected;
    const int32_t* stA = par ? a.state1 : a.state;
    int32_t* stB = par ? a.state : a.state1;
    const int32_t* candA = par ? a.cand1 : a.cand;
    int32_t* candB = par ? a.cand : a.cand1;
    const float4* sA = par ? a.st1 : a.st;
    float4* sB = par ? a.st : a.st1;
    const float* logA = par ? a.log1 : a.cost_log;
    float* logB = par ? a.cost_log : a.log1;
    const int32_t* resA = a.res2 + par * 256; // [4][64]: best delta (bits), its slot, the step's position, its member
    int32_t* resB = a.res2 + (1 - par) * 256;
    // ---- level 1 ----
    const int4 st0 = *reinterpret_cast<const int4*>(stA);
    const int4 st1 = *reinterpret_cast<const int4*>(stA + 4);
    const int4 st2 = *reinterpret_cast<const int4*>(stA + 8);
    const int4 st3 = *reinterpret_cast<const int4*>(stA + 12);
    const float r_delta = __int_as_float(resA[lane]);
    const int r_mm = resA[64 + lane], r_xx = resA[128 + lane], r_x = resA[192 + lane];
    int y_pre[PER];
    float4 s_pre[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int pos = k + tid + 512 * u;
        y_pre[u] = pos < n ? candA[pos] : 0;
        s_pre[u] = pos < n ? sA[pos] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    int med_pre[CLARANS_MAX_MEDOIDS / 64]; // wave 0: the medoids before the swap, slots lane, lane + 64, ...
#pragma unroll
    for (int u = 0; u < CLARANS_MAX_MEDOIDS / 64; ++u) med_pre[u] = (wave == 0 && lane + 64 * u < k) ? candA[lane + 64 * u] : 0;
    const bool cost_wg = b == (int)gridDim.x - 1;
    const int P = st0.x, done = st0.y, log_len = st0.z, rounds = st0.w, fresh = st1.x, err = st1.z;
    const int off = st2.x, stage = st2.y, first = st2.z;
    // the last round of a look also leaves the state block where the host reads it without a copy (mapped host memory)
    int32_t* host = last ? a.host_state : nullptr;
    if (done) { // finished in an earlier round: the state travels on unchanged (both parities stay readable)
        if (b == 0 && tid < 16) {
            stB[tid] = stA[tid];
            if (host) host[tid] = stA[tid];
        }
        return;
    }
    if (cost_wg) { // the running cost: + the addends the previous round's committer logged, in order
        float c = __int_as_float(st1.y);
        if (log_len > 0) {
            c = cost_accumulate<false>(logA, log_len, c, reinterpret_cast<float*>(s_e), reinterpret_cast<float*>(s_we));
        }
        if (tid == 0) {
            stB[ST_COST] = __float_as_int(c);
            if (host) host[ST_COST] = __float_as_int(c);
        }
        return;
    }
    // ---- what the previous round's results mean (every workgroup, the same) ----
    const int W = window_size(corrected, first);
    const int S_prev = fresh ? 0 : stage_size(stage, W - off, a.stage0);
    const unsigned long long neg = __ballot(lane < S_prev && !err && r_delta < 0.0f);
    const bool accept = neg != 0ull;
    const int w = accept ? (int)__builtin_ctzll(neg) : 0;
    const int mm_new = __builtin_amdgcn_readlane(r_mm, w), xx_acc = __builtin_amdgcn_readlane(r_xx, w),
              x_acc = __builtin_amdgcn_readlane(r_x, w);
    int P_n = P, done_n = 0, log_n = 0, rounds_n = rounds, err_n = err, off_n = off, stage_n = stage, first_n = first;
    if (accept) {
        P_n = P + off + w + 1;
        log_n = 1 + n - k;
        rounds_n = rounds + 1;
        off_n = 0;
        stage_n = 0;
        first_n = 0;
        if (P_n + window_size(corrected, 0) > a.draws_len) err_n = 1;
    } else if (!fresh) {
        if (err || off + S_prev >= W) { // error, or `corrected` steps without an accept: this local search is over
            P_n = P + (err ? 0 : W);
            done_n = 1;
        } else {
            off_n = off + S_prev;
            stage_n = stage + 1;
        }
    }
    const int W_n = window_size(corrected, first_n);
    const int S_now = (done_n || err_n) ? 0 : stage_size(stage_n, W_n - off_n, a.stage0);
    const bool committer = b == 0;
    if (!committer && b >= S_now) return; // no step for this workgroup in this round
    // ---- level 2: the accept's gathers, this round's draw ----
    int m_old = 0;
    float d_new[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) d_new[u] = 0.0f;
    if (accept) {
        m_old = candA[mm_new];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int pos = k + tid + 512 * u;
            if (pos < n && pos != xx_acc) d_new[u] = a.D[tri_at(x_acc, y_pre[u])];
        }
    }
    const bool have_step = b < S_now;
    const int xx = have_step ? a.draws[P_n + off_n + b] : k;
    // ---- the accept applied to my positions (clarans_apply_kernel's branches; Clustering.cpp:124-238) ----
    float addend[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) addend[u] = 0.0f;
    float old_dn_xx = 0.0f;
    if (accept) {
        // the position that receives the replaced medoid: distances to the new medoid set, a fresh assignment -- wave 0
        if (wave == 0) {
            float dv[CLARANS_MAX_MEDOIDS / 64];
#pragma unroll
            for (int u = 0; u < CLARANS_MAX_MEDOIDS / 64; ++u) {
                const int mm = lane + 64 * u;
                dv[u] = FLT_MAX;
                if (mm < k) {
                    dv[u] = a.D[tri_at(mm == mm_new ? x_acc : med_pre[u], m_old)];
                    if (committer) a.DMt[(size_t)mm * n + xx_acc] = dv[u];
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
            const float4 s0 = s_pre[u];
            const float dn_y = s0.x, ds_y = s0.y;
            const int an_y = __float_as_int(s0.z), as_y = __float_as_int(s0.w);
            const float dnw = d_new[u];
            float4 out = s0;
            if (an_y == mm_new) { // its medoid is the one that left
                if (dnw < ds_y) {
                    out.x = dnw;
                    addend[u] = __fsub_rn(dnw, dn_y);
                } else {
                    need[u] = true;
                    addend[u] = __fsub_rn(ds_y, dn_y);
                }
            } else if (dnw < dn_y) {
                out = pack_state(dnw, dn_y, mm_new, an_y);
                addend[u] = __fsub_rn(dnw, dn_y);
            } else if (as_y != mm_new && dnw < ds_y) {
                out.y = dnw;
                out.w = __int_as_float(mm_new);
            } else if (as_y != mm_new && dnw > ds_y) {
                // (unchanged: see clarans_apply_kernel)
            } else {
                need[u] = true;
            }
            s_pre[u] = out;
        }
        // The rescans (a few dozen positions per accept, but nearly every wave has one): a lane walking its position's k
        // distances alone waits for k / 8 dependent batches of scattered loads.  Instead the WAVE takes each such position:
        // lane m loads the distance to slot m (+ 64, ...), all of a group's loads in flight together, and the two nearest
        // slots are two wave minima -- "first minimum over the slots, then first minimum over the rest", which is what
        // the sequential scan of Clustering.cpp:262-305 arrives at (see clarans_apply_kernel's last workgroup).
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
    // ---- the committer: the applied state into the other parity ----
    if (committer) {
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int pos = k + tid + 512 * u;
            if (pos >= n) continue;
            candB[pos] = y_pre[u];
            sB[pos] = s_pre[u];
            if (accept) {
                logB[1 + pos - k] = addend[u];
                if (pos == xx_acc) logB[0] = -old_dn_xx;
                else a.DMt[(size_t)mm_new * n + pos] = d_new[u];
            }
        }
        for (int mm = tid; mm < k; mm += 512) candB[mm] = (accept && mm == mm_new) ? x_acc : candA[mm];
        if (tid == 0) {
            stB[ST_P] = P_n;
            stB[ST_DONE] = done_n;
            stB[ST_LOG_LEN] = log_n;
            stB[ST_ROUNDS] = rounds_n;
            stB[ST_FRESH] = 0;
            // (ST_COST: the cost workgroup)
            stB[ST_ERR] = err_n;
            stB[ST_WIN] = 0;
            stB[ST_OFF] = off_n;
            stB[ST_STAGE] = stage_n;
            stB[ST_FIRST] = first_n;
            stB[ST_N_ROUNDS] = st2.w + (fresh ? 0 : 1);
            stB[ST_N_STEPS] = st3.x + (err ? 0 : S_prev);
            stB[ST_N_USEFUL] = st3.y + (accept ? w + 1 : (err ? 0 : S_prev));
            stB[ST_N_COMMON] = st3.z;
            stB[ST_N_GENERAL] = st3.w;
            if (host) { // (every word but the cost, which the cost workgroup leaves there)
                host[ST_P] = P_n; host[ST_DONE] = done_n; host[ST_LOG_LEN] = log_n; host[ST_ROUNDS] = rounds_n; host[ST_FRESH] = 0;
                host[ST_ERR] = err_n; host[ST_WIN] = 0; host[ST_OFF] = off_n; host[ST_STAGE] = stage_n; host[ST_FIRST] = first_n;
                host[ST_N_ROUNDS] = st2.w + (fresh ? 0 : 1); host[ST_N_STEPS] = st3.x + (err ? 0 : S_prev);
                host[ST_N_USEFUL] = st3.y + (accept ? w + 1 : (err ? 0 : S_prev)); host[ST_N_COMMON] = st3.z; host[ST_N_GENERAL] = st3.w;
            }
        }
    }
    if (!have_step) return;
    // ---- level 3 / 4: my step of this round against the applied state ----
    const int x = (accept && xx == xx_acc) ? m_old : candA[xx];
    float best = 0.0f;
    int bk = INT_MAX;
    __syncthreads(); // (s_xx_state has been consumed; evaluate_step stages through LDS)
    evaluate_step<KPT, false, true, false>(a, xx, x, y_pre, s_pre, s_e, s_we, best, bk, nullptr);
    if (tid == 0) {
        resB[b] = __float_as_int(best);
        resB[64 + b] = bk;
        resB[128 + b] = xx;
        resB[192 + b] = x;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The rounds of a search inside ONE launch, kept on ONE XCD (opt-in, LCSGPU_CLARANS_CHAIN=1).
//
// A round is two dependent steps (evaluate the stage's pending steps; apply the first improving one), and as two launches
// it pays two kernel boundaries and, after each, a trip to memory per dependent load level -- a boundary leaves nothing in
// the caches that another XCD wrote.  Workgroups that all run on the SAME XCD share one L2: a barrier among them is an
// atomic at that L2 (about 1 us for 17 workgroups, scripts/ubench_xcd.hip), a store is visible to the others once it has
// left the CU (s_waitcnt vmcnt(0); the vector L1 is write-through) and a reader only has to bypass its own L1 -- no fence
// wider than the workgroup.  So: every array another workgroup of the search writes during the launch (cand, st, DMt, the
// windows, the step results, the cost log) is read with sc1 loads (ldc below); the distances D are read-only and take
// the ordinary path.  The workgroups are picked by where they really run: each reads XCC_ID and takes a ticket of a
// search assigned to that XCD; the first P are that search's ranks, everything else exits at once.  Ranks 0 .. P-2
// evaluate the steps of a stage (step b -> rank b mod (P-1)) and apply an accepted step to their share of the positions;
// rank P-1 keeps the running cost and does the apply kernel's last-workgroup part.  The control state (next draw, window,
// stage, ...) is computed by every workgroup from the same step results, so a round needs exactly two barriers.
// Arithmetic, comparison directions and the order of every float addition are those of the two kernels above.
enum { CH_TICKET = 16, CH_GO = 17, CH_BAR = 32 }; // words of the search's 64-word state block (zeroed by the host before every launch)

__device__ __forceinline__ unsigned xcc_id()
{
    // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20), offset 0, size 32
    return (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu;
}
// all P ranks of a search arrive; nothing but the workgroup's own stores having left the CU is waited for
__device__ __forceinline__ bool chain_barrier(int* counter, int target)
{
    __shared__ int s_ok;
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0, ok = 1;
        while (ldc(counter) < target)
            if (++spins > (1 << 20)) { ok = 0; break; } // ~1 s: a rank of this search is gone
        s_ok = ok;
    }
    __syncthreads();
    return s_ok != 0;
}

template <int KPT>
__global__ __launch_bounds__(512) void clarans_chain_kernel(ClaransBatch batch, int P, int max_rounds)
{
    __shared__ float4 s_e[1024];    // 16 KB
    __shared__ float4 s_we[8][128]; // 16 KB
    __shared__ int s_rank, s_search;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) {
        int rank = -1, search = -1;
        for (int i = (int)xcc_id(); i < batch.n; i += 8) { // the searches assigned to the XCD this workgroup runs on
            const int t = atomicAdd(&batch.s[i].state[CH_TICKET], 1);
            if (t < P) {
                rank = t;
                search = i;
                break;
            }
        }
        s_rank = rank;
        s_search = search;
    }
    __syncthreads();
    const int rank = s_rank;
    if (rank < 0) return;
    const ClaransArgs& a = batch.s[s_search];
    int* st = a.state;
    // assembly: rank 0 decides whether all P ranks have found a seat on this XCD; if not, nothing has been touched
    // and the host runs this look as ordinary rounds
    if (tid == 0) {
        if (rank == 0) {
            int spins = 0, go = 1;
            while (ldc(&st[CH_TICKET]) < P) {
                if (++spins > (1 << 11)) { go = 2; break; } // ~3 ms: this XCD has no room for the search right now
                __builtin_amdgcn_s_sleep(8);
            }
            __hip_atomic_store(&st[CH_GO], go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (ldc(&st[CH_GO]) == 0) __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    if (ldc(&st[CH_GO]) != 1) return;

    const int corrected = a.corrected, k = a.n_medoids, n = a.n_elems;
    int p = st[ST_P], done = st[ST_DONE], log_len = st[ST_LOG_LEN], accepts = st[ST_ROUNDS], err = st[ST_ERR], win = st[ST_WIN] & 1,
        off = st[ST_OFF], stage = st[ST_STAGE], first = st[ST_FIRST];
    float cost = __int_as_float(st[ST_COST]);
    const bool tail = rank == P - 1;
    const int n_eval = P - 1;
    const int W_next = window_size(corrected, 0);
    int bar = 0;
    // where the time of a round goes (s_memtime ticks of 10 ns), ranks 0 and P-1: evaluate, wait, apply, wait
    unsigned long long t_ph[4] = {0, 0, 0, 0}, t0 = wall_clock64();
    int n_rounds = 0;
    const unsigned long long clk0 = __builtin_readcyclecounter(), wall0 = wall_clock64(); // shader clocks per 10 ns tick = the clock the kernel ran at
    unsigned long long t_ev[5] = {0, 0, 0, 0, 0}; // inside an evaluation: loads, staging, own walk, slowest wave, reduction
    auto lap = [&](int ph) {
        const unsigned long long t1 = wall_clock64();
        t_ph[ph] += t1 - t0;
        t0 = t1;
    };
    for (int r = 0; r < max_rounds && !done; ++r) {
        const int W = window_size(corrected, first);
        const int S = stage_size(stage, W - off, a.stage0);
        ++n_rounds;
        // ---- evaluate
        if (tail) {
            if (log_len) cost = cost_accumulate<true>(a.cost_log, log_len, cost, reinterpret_cast<float*>(s_e), reinterpret_cast<float*>(s_we));
        } else if (!err) {
            for (int b = rank; b < S; b += n_eval) {
                const int xx = ldc(&a.win_xx[win * a.win_cap + off + b]);
                const int x = ldc(&a.win_x[win * a.win_cap + off + b]);
                float best = 0.0f;
                int bk = INT_MAX;
                int y_pre[4];
                float4 s_pre[4];
                evaluate_step<KPT, true, false, true>(a, xx, x, y_pre, s_pre, s_e, s_we, best, bk, t_ev);
                __syncthreads(); // the staging areas are free for the next step
                if (tid == 0) {
                    a.res_delta[b] = best;
                    a.res_mm[b] = bk;
                }
            }
        }
        lap(0);
        bar += P;
        if (!chain_barrier(&st[CH_BAR], bar)) {
            if (tid == 0) atomicExch(&st[ST_ERR], 2);
            return;
        }
        lap(1);
        // ---- apply: the first step of the stage with a negative delta
        const bool in_stage = !err && lane < S;
        const float rd = in_stage ? ldc(&a.res_delta[lane]) : 0.0f;
        const uint64_t neg = __ballot(in_stage && rd < 0.0f);
        const bool accept = neg != 0;
        const int w = accept ? __ffsll((unsigned long long)neg) - 1 : 0;
        const int j = off + w;
        if (accept) {
            const int xx = ldc(&a.win_xx[win * a.win_cap + j]);
            const int x = ldc(&a.win_x[win * a.win_cap + j]); // the new medoid
            const int mm_new = ldc(&a.res_mm[w]);
            if (tail) {
                const int p_new = p + j + 1;
                if (tid < 64) {
                    const int m_old = ldc(&a.cand[mm_new]); // the medoid that is replaced; from now on it sits at position xx
                    if (p_new + W_next <= a.draws_len) {
                        int32_t* nxx = a.win_xx + (1 - win) * a.win_cap;
                        int32_t* nx = a.win_x + (1 - win) * a.win_cap;
                        for (int q = tid; q < W_next; q += 64) {
                            const int xn = a.draws[p_new + q];
                            nxx[q] = xn;
                            nx[q] = xn == xx ? m_old : ldc(&a.cand[xn]); // draws are non-medoid positions
                        }
                    }
                    const float old_dn = ldc4(&a.st[xx]).x;
                    float dv[CLARANS_MAX_MEDOIDS / 64];
#pragma unroll
                    for (int u = 0; u < CLARANS_MAX_MEDOIDS / 64; ++u) {
                        const int mm = tid + 64 * u;
                        dv[u] = FLT_MAX;
                        if (mm < k) {
                            dv[u] = a.D[tri_at(mm == mm_new ? x : ldc(&a.cand[mm]), m_old)];
                            a.DMt[(size_t)mm * n + xx] = dv[u];
                        }
                    }
                    float v1 = FLT_MAX, v2 = FLT_MAX;
                    int i1 = INT_MAX, i2 = INT_MAX;
#pragma unroll
                    for (int u = 0; u < CLARANS_MAX_MEDOIDS / 64; ++u) {
                        const int mm = tid + 64 * u;
                        if (mm < k && (dv[u] < v1 || i1 == INT_MAX)) { v1 = dv[u]; i1 = mm; }
                    }
                    wave_first_min_valid(v1, i1);
#pragma unroll
                    for (int u = 0; u < CLARANS_MAX_MEDOIDS / 64; ++u) {
                        const int mm = tid + 64 * u;
                        if (mm < k && mm != i1 && (dv[u] < v2 || i2 == INT_MAX)) { v2 = dv[u]; i2 = mm; }
                    }
                    wave_first_min_valid(v2, i2);
                    if (tid == 0) {
                        const bool has1 = i1 != INT_MAX && v1 < FLT_MAX, has2 = i2 != INT_MAX && v2 < FLT_MAX;
                        a.st[xx] = pack_state(has1 ? v1 : FLT_MAX, has2 ? v2 : FLT_MAX, has1 ? i1 : -1, has2 ? i2 : -1);
                        a.cost_log[0] = -old_dn;
                        a.cost_log[1 + xx - k] = has1 ? v1 : FLT_MAX;
                        // the swap itself: nobody else reads cand between the two barriers of a round
                        a.cand[mm_new] = x;
                        a.cand[xx] = m_old;
                    }
                }
            } else {
                for (int yy = k + rank * 512 + tid; yy < n; yy += n_eval * 512) {
                    if (yy == xx) continue;
                    const int y_mine = ldc(&a.cand[yy]);
                    const float4 s = ldc4(&a.st[yy]);
                    const float d_new = a.D[tri_at(x, y_mine)];
                    a.DMt[(size_t)mm_new * n + yy] = d_new;
                    const float dn_y = s.x, ds_y = s.y;
                    const int an_y = __float_as_int(s.z), as_y = __float_as_int(s.w);
                    float addend = 0.0f;
                    bool rescan = false;
                    float4 out = s;
                    if (an_y == mm_new) { // its medoid is the one that left
                        if (d_new < ds_y) {
                            out.x = d_new;
                            addend = __fsub_rn(d_new, dn_y);
                        } else {
                            rescan = true;
                            addend = __fsub_rn(ds_y, dn_y);
                        }
                    } else if (d_new < dn_y) {
                        out = pack_state(d_new, dn_y, mm_new, an_y);
                        addend = __fsub_rn(d_new, dn_y);
                    } else if (as_y != mm_new && d_new < ds_y) {
                        out.y = d_new;
                        out.w = __int_as_float(mm_new);
                    } else {
                        rescan = true;
                    }
                    if (rescan) { // CLARANS::updateAssignment over the new medoid set
                        Nearest2 nb;
                        const float* col = a.DMt + yy;
                        for (int m0 = 0; m0 < k; m0 += 8) {
                            float v[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) v[u] = ldc(&col[(size_t)min(m0 + u, k - 1) * n]);
#pragma unroll
                            for (int u = 0; u < 8; ++u)
                                if (m0 + u < k) nb.feed(m0 + u == mm_new ? d_new : v[u], m0 + u);
                        }
                        out = pack_state(nb.dn, nb.ds, nb.an, nb.as);
                    }
                    a.st[yy] = out;
                    a.cost_log[1 + yy - k] = addend;
                }
            }
        }
        lap(2);
        bar += P;
        if (!chain_barrier(&st[CH_BAR], bar)) {
            if (tid == 0) atomicExch(&st[ST_ERR], 2);
            return;
        }
        lap(3);
        // ---- the control state, the same in every workgroup
        if (accept) {
            const int p_new = p + j + 1;
            if (p_new + W_next > a.draws_len) err = 1;
            p = p_new;
            log_len = 1 + n - k;
            ++accepts;
            win = 1 - win;
            off = 0;
            stage = 0;
            first = 0;
        } else {
            log_len = 0;
            if (err || off + S >= W) { // error, or `corrected` steps without an accept: this local search is over
                p = p + (err ? 0 : W);
                done = 1;
            } else {
                off += S;
                ++stage;
            }
        }
    }
    if (tail && tid == 0) {
        st[ST_P] = p;
        st[ST_DONE] = done;
        st[ST_LOG_LEN] = log_len;
        st[ST_ROUNDS] = accepts;
        st[ST_COST] = __float_as_int(cost);
        st[ST_ERR] = err;
        st[ST_WIN] = win;
        st[ST_OFF] = off;
        st[ST_STAGE] = stage;
        st[ST_FIRST] = first;
    }
    if ((tail || rank == 0) && tid == 0) { // words 48 .. 57: LCSGPU_PROFILE
        int* dbg = st + (tail ? 53 : 48);
        dbg[0] = n_rounds;
        for (int q = 0; q < 4; ++q) dbg[1 + q] = (int)(t_ph[q] & 0x7fffffff);
        if (!tail) {
            for (int q = 0; q < 5; ++q) st[58 + q] = (int)(t_ev[q] & 0x7fffffff);
            const unsigned long long dw = wall_clock64() - wall0;
            st[63] = dw ? (int)((__builtin_readcyclecounter() - clk0) * 100 / dw) : 0; // MHz
        }
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

// `rounds` x (evaluate the next stage of every search's window, apply).  The first window of a local
// search has `corrected` steps, the later ones corrected - 1 (the reference resets its step counter
// to 1 after an accept).
// `rounds` launches of clarans_round_kernel (an even number: the host reads the parity-0 buffers)
hipError_t launch_clarans_rounds_fused(const ClaransBatch& b, int rounds, hipStream_t stream)
{
    int kpt = 1, steps = 0;
    for (int i = 0; i < b.n; ++i) {
        const ClaransArgs& a = b.s[i];
        kpt = std::max(kpt, ((a.n_medoids + 7) / 8 + 63) / 64);
        steps = std::max(steps, std::max(1, std::min(a.corrected, STAGE_MAX)));
    }
    const dim3 grid(steps + 1, b.n), block(512); // a stage has at most STAGE_MAX steps; + the cost workgroup
    for (int r = 0; r < rounds; ++r) {
        const int last = r == rounds - 1 ? 1 : 0;
        if (kpt <= 1) hipLaunchKernelGGL(clarans_round_kernel<1>, grid, block, 0, stream, b, r & 1, last);
        else hipLaunchKernelGGL(clarans_round_kernel<2>, grid, block, 0, stream, b, r & 1, last);
    }
    return hipGetLastError();
}

hipError_t clarans_lists_ticks(unsigned long long out[8])
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lists_ticks), 64, 0, hipMemcpyDeviceToHost);
}

hipError_t launch_clarans_rounds(const ClaransBatch& b, int rounds, hipStream_t stream)
{
    int kpt = 1, apply_blocks = 1, steps = 0;
    for (int i = 0; i < b.n; ++i) {
        const ClaransArgs& a = b.s[i];
        kpt = std::max(kpt, ((a.n_medoids + 7) / 8 + 63) / 64); // slots per lane: 8 waves share the slots
        apply_blocks = std::max(apply_blocks, (a.n_elems - a.n_medoids + 63) / 64 + 1);
        steps = std::max(steps, std::min(a.corrected, STAGE_MAX));
    }
    const dim3 grid(steps + 1, b.n), block(512); // a stage has at most STAGE_MAX steps; + the cost workgroup
    for (int r = 0; r < rounds; ++r) {
        if (kpt <= 1) hipLaunchKernelGGL(clarans_eval_kernel<1>, grid, block, 0, stream, b);
        else hipLaunchKernelGGL(clarans_eval_kernel<2>, grid, block, 0, stream, b);
        hipLaunchKernelGGL(clarans_apply_kernel, dim3(apply_blocks, b.n), dim3(64), 0, stream, b);
    }
    return hipGetLastError();
}

// Up to `rounds` rounds of every search of the batch inside one launch (LCSGPU_CLARANS_CHAIN=1): search i runs on XCD i mod 8
// with `ranks` workgroups.  The caller has zeroed words 16 .. 47 of every search's state block on the same stream.
hipError_t launch_clarans_chain(const ClaransBatch& b, int rounds, int ranks, hipStream_t stream)
{
    int kpt = 1;
    for (int i = 0; i < b.n; ++i) kpt = std::max(kpt, ((b.s[i].n_medoids + 7) / 8 + 63) / 64);
    const int per_xcd = (b.n + 7) / 8;
    const dim3 grid(8 * (per_xcd * ranks + 16)), block(512); // XCDs get workgroups round robin: enough for every XCD to seat its searches
    if (kpt <= 1) hipLaunchKernelGGL(clarans_chain_kernel<1>, grid, block, 0, stream, b, ranks, rounds);
    else hipLaunchKernelGGL(clarans_chain_kernel<2>, grid, block, 0, stream, b, ranks, rounds);
    return hipGetLastError();
}

} // namespace lcsgpu
