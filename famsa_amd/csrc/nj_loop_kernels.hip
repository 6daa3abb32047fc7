// nj_loop_kernels.hip -- neighbour joining as ONE resident launch: every merge of
// NeighborJoining::computeTree (reference tree/NeighborJoining.cpp:33-118) inside a loop on the
// device, the workgroups exchanging a few words through memory instead of meeting at four kernel
// boundaries per merge.
//
// The arithmetic is tree_kernels.hip's nj_*_kernel operation for operation (the reference's float
// association, its sequential sums in ascending cluster order, the first strict minimum of q in
// (i, j) lexicographic order); what changes is where the work waits.  Every workgroup OWNS a
// share of the flat triangle (every G-th block of 1 KB) -- it alone reads those distances with
// ordinary loads and it alone rewrites them -- and keeps the n cluster sums and active flags as
// copies in its LDS.
// Per merge:
//
//   1  scan the own share for the smallest q (16-byte loads; a row that is gone has NaN for its sum
//      and loses by itself); the last wave to finish posts the workgroup's candidate (two tagged 64-bit
//      words) and polls everybody's -- WHILE wave 0 adds up the merged cluster's sum of the previous
//      merge in ascending order, rounding as the reference's one-after-the-other float additions do
//      (ordered_sum.h): the chain runs beside the scan AND the exchange, in every workgroup's own copy.
//   2  the pairs of the merged cluster (its distances are the chain's addends, in LDS) are the same in
//      every workgroup and need no exchange: their smallest q against the exchanged one -> (mi, mj).
//   3  for the entries (mi, k) of the own share: u = Dik + Djk, post u tagged, store the new distance.
//   4  poll the u of all active k; everybody derives the same new distances (u - Dij) / 2 and the
//      same sums (sum - u) + d from them in its own LDS.
//
// What crosses workgroups goes through agent-scope atomic loads and stores (write-through, cache-
// bypassing: the eight XCDs' L2s are not coherent with each other) and carries its merge number as
// a tag, so there is no read-modify-write and no cache-wide release / acquire on the path: a
// counter barrier of 256 workgroups costs 7-11 us here, an exchange of tagged slots 3.3 us, and the
// fences add 5 us (scripts/ubench_gridbar.hip, profiles/gridbar_r06.txt).  Whenever an eighth of the
// rows has been merged away the triangle is rewritten without them (order kept: ties are decided by
// positions, and positions stay in ascending order), with real fences around it, so a scan reads
// what is alive: n^3/3 x 2 B in total instead of n^3 x 2 B.
//
// One workgroup per CU, so that all are resident at once; a word that does not arrive within ~1 s
// sets an error flag and every workgroup leaves (lcsgpu_nj then runs the merges as launches).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nj_loop.h"
#include "ordered_sum.h"
#include "dpp_min.h"

namespace lcsgpu {

namespace {

constexpr int NJ_THREADS = 1024;
constexpr float NJ_QMAX = 3.40282347e38f; // numeric_limits<float>::max(): only q < max can win
constexpr uint32_t NJ_SPIN_LIMIT = 1u << 20;
constexpr int NJ_NONE = 0xffff;           // "no candidate" in a posted slot (row ids are below 16384)

#define NJ_AGENT __HIP_MEMORY_SCOPE_AGENT

struct Cand {
    float q;
    int i, j;
};

// the reference's scan order is i outer, j inner with a strict "<": among equal q the smallest (i, j) wins
__device__ __forceinline__ bool cand_less(float q2, int i2, int j2, const Cand& c)
{
    return q2 < c.q || (q2 == c.q && (i2 < c.i || (i2 == c.i && j2 < c.j)));
}

__device__ __forceinline__ void cand_take(Cand& c, float q, int i, int j)
{
    if (q < NJ_QMAX && cand_less(q, i, j, c)) { c.q = q; c.i = i; c.j = j; }
}

// the wave's smallest candidate in every lane: dpp_min.h's (value, index) first minimum with (i, j) as one index
// (i and j are below 2^14; "none" packs to the largest index and has the largest q)
__device__ __forceinline__ Cand wave_min(Cand c)
{
    float q = c.q;
    uint32_t ij = c.i == 0x7fffffff ? 0xffffffffu : ((uint32_t)c.i << 16 | (uint32_t)c.j);
    wave_first_min(q, ij);
    Cand r;
    r.q = q;
    r.i = ij == 0xffffffffu ? 0x7fffffff : (int)(ij >> 16);
    r.j = ij == 0xffffffffu ? 0x7fffffff : (int)(ij & 0xffff);
    return r;
}

// all threads get the workgroup's minimum; scratch: 16 candidates
__device__ __forceinline__ Cand block_min(Cand c, Cand* scratch)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    c = wave_min(c);
    __syncthreads(); // scratch may still be read from the previous use
    if (lane == 0) scratch[wave] = c;
    __syncthreads();
    return wave_min(scratch[lane & 15]);
}

__device__ __forceinline__ size_t tri(uint32_t i, uint32_t j) // TriangleMatrix::access
{
    return i >= j ? j + (size_t)i * (i - 1) / 2 : i + (size_t)j * (j - 1) / 2;
}

// row of flat index e: j with j(j-1)/2 <= e < j(j+1)/2   (e < 2^27: n <= 16384)
__device__ __forceinline__ int row_of(int e)
{
    int j = (int)((1.0f + __fsqrt_rn(1.0f + 8.0f * (float)e)) * 0.5f);
    while (j * (j - 1) / 2 > e) --j;
    while (j * (j + 1) / 2 <= e) ++j;
    return j;
}

// what crosses workgroups: write-through stores, cache-bypassing loads
__device__ __forceinline__ uint64_t ld_u64(const uint64_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, NJ_AGENT); }
__device__ __forceinline__ void st_u64(uint64_t* p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, NJ_AGENT); }
__device__ __forceinline__ float ld_f32(const float* p)
{
    return __uint_as_float(__hip_atomic_load((const uint32_t*)p, __ATOMIC_RELAXED, NJ_AGENT));
}
__device__ __forceinline__ void st_f32(float* p, float v)
{
    __hip_atomic_store((uint32_t*)p, __float_as_uint(v), __ATOMIC_RELAXED, NJ_AGENT);
}

struct Shared {
    float* sum;     // [cap] the clusters' sums; NaN: the row is gone (or its sum is still being added up)
    float* tmp;     // [cap] the new distances of the last merge (the addends of the chain); the map of a squeeze
    uint8_t* act;   // [cap]
    Cand* cand;     // [16] the waves' candidates  [16] everybody's smallest
    int* misc;      // [0] abort  [1] the chain's result  [2] scanning waves that have finished
    int* scan;      // [1024]
};

// Post two words tagged with `epoch`, wait for everybody's.  Thread t < G returns with workgroup t's words in
// (r0, r1).  The waves' earlier stores have been acknowledged before the post (s_waitcnt: they are write-through), so
// whoever sees the post sees them.  false: somebody did not arrive.
__device__ __forceinline__ bool exchange(const NjLoopArgs& p, const Shared& S, uint32_t epoch, uint32_t lo0, uint32_t lo1,
                                         uint64_t& r0, uint64_t& r1)
{
    const int G = gridDim.x, w = blockIdx.x, tid = threadIdx.x;
    uint64_t* buf = p.slots + (size_t)(epoch & 1) * 2 * G;
    const uint64_t tag = (uint64_t)epoch << 32;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        st_u64(buf + 2 * w, tag | lo0);
        st_u64(buf + 2 * w + 1, tag | lo1);
    }
    r0 = r1 = 0;
    for (uint32_t spins = 0;; ++spins) {
        int ok = 1;
        if (tid < G) {
            r0 = ld_u64(buf + 2 * tid);
            r1 = ld_u64(buf + 2 * tid + 1);
            ok = (r0 >> 32) == epoch && (r1 >> 32) == epoch;
        }
        if (__syncthreads_and(ok)) return true;
        if (spins > NJ_SPIN_LIMIT) {
            if (tid == 0) __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, NJ_AGENT);
            return false;
        }
    }
}

} // namespace

__global__ __launch_bounds__(NJ_THREADS) void nj_loop_kernel(NjLoopArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int cap = p.cap; // >= ordered_sum_padded(n), a multiple of 16
    Shared S;
    S.sum = (float*)smem;
    S.tmp = S.sum + cap;
    S.act = (uint8_t*)(S.tmp + cap);
    S.cand = (Cand*)(S.act + cap);
    S.misc = (int*)(S.cand + 20);
    S.scan = S.misc + 4;

    const NjArgs& a = p.a;
    const int G = gridDim.x, w = blockIdx.x, tid = threadIdx.x;
    const float nan = __builtin_nanf("");
    float* D = a.D;
    float* Dalt = p.D2;
    int n_cur = a.n, n_act = a.n;
    uint32_t epoch = 0;
    int mi = -1;                // the row whose sum the chain still owes (chain_pending)
    bool chain_pending = false;

    if (tid == 0) S.misc[0] = S.misc[2] = 0;
    for (int t = tid; t < n_cur; t += NJ_THREADS) {
        S.sum[t] = a.sum[t];
        S.act[t] = 1;
    }
    __syncthreads();

    long long lap[6] = {0, 0, 0, 0, 0, 0}, t_last = wall_clock64(); // workgroup 0's account (10 ns ticks)
#define NJ_LAP(x) { const long long t_now = wall_clock64(); lap[x] += t_now - t_last; t_last = t_now; }
    int iter = 0;
    for (; n_act > 2; ++iter) {
        // ---------------- the triangle without the rows that are gone ----------------
        if (n_cur - n_act >= max(n_cur >> 3, 16) && n_cur >= p.compact_min) {
            uint64_t r0, r1;
            if (chain_pending) {
                if (tid < 64) {
                    const float sm = wave_ordered_sum(S.tmp, n_cur);
                    if (tid == 0) S.sum[mi] = sm;
                }
                chain_pending = false;
            }
            // the stores of the last merge have landed everywhere (they are read below through the cache-bypassing path)
            if (!exchange(p, S, ++epoch, 0, 0, r0, r1)) return;
            // new -> old map (ascending): an exclusive scan of the active flags, <= 16 rows a thread
            int* s_map = (int*)S.tmp;
            const int per = (n_cur + NJ_THREADS - 1) / NJ_THREADS;
            const int t0 = min(n_cur, tid * per), t1 = min(n_cur, t0 + per);
            int cnt = 0;
            for (int t = t0; t < t1; ++t) cnt += S.act[t];
            S.scan[tid] = cnt;
            __syncthreads();
            for (int s = 1; s < NJ_THREADS; s <<= 1) { // Hillis-Steele over 1024 counts
                const int add = tid >= s ? S.scan[tid - s] : 0;
                __syncthreads();
                S.scan[tid] += add;
                __syncthreads();
            }
            int pos = S.scan[tid] - cnt;
            for (int t = t0; t < t1; ++t)
                if (S.act[t]) s_map[pos++] = t;
            __syncthreads();
            const int n_new = n_act;
            {
                const int T = n_new * (n_new - 1) / 2, T4 = (T + 3) >> 2, NB = (T4 + 63) >> 6;
                float4* O4 = (float4*)Dalt;
                for (int b = w + G * (tid >> 6); b < NB; b += G * 16) {
                    const int g = (b << 6) + (tid & 63);
                    if (g >= T4) continue;
                    const int e = g << 2;
                    int j = row_of(e), i = e - j * (j - 1) / 2;
                    float x[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        x[c] = 0.0f;
                        if (e + c < T) {
                            const uint32_t jo = s_map[j], io = s_map[i]; // jo > io
                            x[c] = ld_f32(D + (size_t)jo * (jo - 1) / 2 + io);
                        }
                        if (++i == j) { i = 0; ++j; }
                    }
                    O4[g] = float4{x[0], x[1], x[2], x[3]};
                }
            }
            // sums (every workgroup's copy) and node ids (workgroup 0's) move with their rows
            {
                float keep[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int t = tid + u * NJ_THREADS;
                    keep[u] = t < n_new ? S.sum[s_map[t]] : 0.0f;
                }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int t = tid + u * NJ_THREADS;
                    if (t < n_new) {
                        S.sum[t] = keep[u];
                        S.act[t] = 1;
                    }
                }
            }
            if (w == 0) {
                int keep[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int t = tid + u * NJ_THREADS;
                    keep[u] = t < n_new ? a.node[s_map[t]] : 0;
                }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const int t = tid + u * NJ_THREADS;
                    if (t < n_new) a.node[t] = keep[u];
                }
            }
            n_cur = n_new;
            float* sw = D; D = Dalt; Dalt = sw;
            // the new triangle was written with ordinary stores: out of this L2 before anybody is told
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (!exchange(p, S, ++epoch, 0, 0, r0, r1)) return;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            NJ_LAP(0)
        }

        const float f = (float)(n_act - 2);
        const int T = n_cur * (n_cur - 1) / 2, T4 = (T + 3) >> 2, NB = (T4 + 63) >> 6; // blocks of 64 x 16 bytes

        // ---------------- 1 + 2: the own share's smallest q, everybody's candidates  ||  the chain of the previous merge ----------------
        // Wave 0 adds up the sum the previous merge owes; the other waves scan, and the LAST of them to finish posts the
        // workgroup's candidate and polls everybody's -- the chain runs beside the scan AND the exchange, and the
        // workgroup meets once both are done.
        int mj;
        {
            const int wave = tid >> 6, lane = tid & 63;
            const int first_wave = chain_pending ? 1 : 0, nw = 16 - first_wave;
            ++epoch; // this merge's number: the tag of its candidates and of its u
            if (wave < first_wave) {
                // ci.sum = the new distances added up in ascending cluster order (NeighborJoining.cpp:88-108), rounded as the
                // reference's one-after-the-other additions (ordered_sum.h)
                const long long c0 = wall_clock64();
                const float sm = wave_ordered_sum(S.tmp, n_cur);
                if (tid == 0) {
                    S.misc[1] = __float_as_int(sm);
                    lap[5] += wall_clock64() - c0;
                }
            } else {
                Cand best{NJ_QMAX, 0x7fffffff, 0x7fffffff};
                const int sw = wave - first_wave;
                const float4* D4 = (const float4*)D;
                for (int bb = w + G * sw; bb < NB; bb += 4 * G * nw) {
                    float4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int g = ((bb + u * G * nw) << 6) + lane;
                        v[u] = g < T4 ? D4[g] : float4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int g = ((bb + u * G * nw) << 6) + lane;
                        if (g >= T4) break;
                        const int e = g << 2;
                        int j = row_of(e), i = e - j * (j - 1) / 2;
                        const float x[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                        float sj = S.sum[j];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            if (e + c < T) { // NaN sums (rows that are gone, the row the chain owes) make q NaN: it loses
                                const float q = __fsub_rn(__fsub_rn(__fmul_rn(f, x[c]), S.sum[i]), sj);
                                cand_take(best, q, i, j);
                            }
                            if (++i == j) { i = 0; ++j; sj = S.sum[j]; }
                        }
                    }
                }
                best = wave_min(best);
                int arrived = 0;
                if (lane == 0) {
                    S.cand[wave] = best;
                    arrived = __hip_atomic_fetch_add(&S.misc[2], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                arrived = __builtin_amdgcn_readfirstlane(arrived);
                if (arrived == nw - 1) { // the last scanning wave: the workgroup's candidate out, everybody's in
                    Cand c{NJ_QMAX, 0x7fffffff, 0x7fffffff};
                    if (lane >= first_wave && lane < 16) c = S.cand[lane];
                    c = wave_min(c);
                    uint64_t* buf = p.slots + (size_t)(epoch & 1) * 2 * G;
                    const uint64_t tag = (uint64_t)epoch << 32;
                    if (lane == 0) {
                        const bool none = c.i == 0x7fffffff;
                        st_u64(buf + 2 * w, tag | __float_as_uint(c.q));
                        st_u64(buf + 2 * w + 1, tag | (none ? ((uint32_t)NJ_NONE << 16 | NJ_NONE) : ((uint32_t)c.i << 16 | (uint32_t)c.j)));
                    }
                    Cand g{NJ_QMAX, 0x7fffffff, 0x7fffffff};
                    bool lost = false;
                    for (int t = lane; t < G; t += 64) {
                        uint64_t r0 = 0, r1 = 0;
                        for (uint32_t spins = 0;; ++spins) {
                            if ((r0 >> 32) != epoch) r0 = ld_u64(buf + 2 * t);
                            if ((r1 >> 32) != epoch) r1 = ld_u64(buf + 2 * t + 1);
                            if ((r0 >> 32) == epoch && (r1 >> 32) == epoch) break;
                            if (spins > NJ_SPIN_LIMIT) { lost = true; break; }
                        }
                        const int i = (int)((uint32_t)r1 >> 16), j = (int)(r1 & 0xffff);
                        if (i != NJ_NONE && cand_less(__uint_as_float((uint32_t)r0), i, j, g)) {
                            g.q = __uint_as_float((uint32_t)r0); g.i = i; g.j = j;
                        }
                    }
                    g = wave_min(g);
                    if (lane == 0) {
                        S.cand[16] = g;
                        S.misc[2] = 0;
                        if (lost) {
                            S.misc[0] = 1;
                            __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, NJ_AGENT);
                        }
                    }
                }
            }
            __syncthreads();
            if (S.misc[0]) return;
            Cand c = S.cand[16];
            if (chain_pending) {
                // the merged cluster's pairs: its sum is the chain's result, its distances are the chain's addends -- the same
                // in every workgroup, so they need no exchange
                const float si = __int_as_float(S.misc[1]);
                if (tid == 0) S.sum[mi] = si; // (read again after block_min's barriers only)
                Cand best{NJ_QMAX, 0x7fffffff, 0x7fffffff};
                for (int k = tid; k < n_cur; k += NJ_THREADS) {
                    if (k == mi) continue;
                    const float d = __fmul_rn(f, S.tmp[k]);
                    const float q = k < mi ? __fsub_rn(__fsub_rn(d, S.sum[k]), si) : __fsub_rn(__fsub_rn(d, si), S.sum[k]);
                    cand_take(best, q, min(k, mi), max(k, mi));
                }
                best = block_min(best, S.cand);
                if (cand_less(best.q, best.i, best.j, c)) c = best;
                chain_pending = false;
            }
            if (c.i == 0x7fffffff) { // no q below FLT_MAX: the reference's result is degenerate (min_i = min_j = 0)
                if (w == 0 && tid == 0) a.sel[2] = 1;
                return;
            }
            mi = c.i;
            mj = c.j;
        }
        NJ_LAP(1)
        if (w == 0 && tid == 0) {
            a.left[iter] = a.node[mi];
            a.right[iter] = a.node[mj];
            a.node[mi] = a.n + iter;
        }

        // ---------------- 3: the entries (mi, k) of the own share ----------------
        const float Dij = ld_f32(D + tri(mi, mj));
        const uint64_t tag = (uint64_t)epoch << 32;
        {
            auto update = [&](int k, int flat) {
                if (!S.act[k] || k == mj) return;
                const float Dik = D[flat], Djk = ld_f32(D + tri(mj, k));
                const float u = __fadd_rn(Dik, Djk);
                st_u64(p.u + k, tag | __float_as_uint(u));
                st_f32(D + flat, __fmul_rn(__fsub_rn(u, Dij), 0.5f)); // (Dik + Djk - Dij) / 2
            };
            for (int k = tid; k < n_cur; k += NJ_THREADS) {
                if (k == mi) continue;
                const int flat = (int)tri(mi, k);
                if (((flat >> 8) % G) == w) update(k, flat); // the block of 256 floats it lies in is this workgroup's
            }
        }

        NJ_LAP(3)
        // ---------------- 4: everybody's u -> the new distances and the sums, in LDS ----------------
        for (int kb = tid; kb < n_cur; kb += 4 * NJ_THREADS) {
            uint64_t v[4];
            bool need[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = kb + u * NJ_THREADS;
                need[u] = k < n_cur && S.act[k] && k != mi && k != mj;
                v[u] = 0;
            }
            for (uint32_t spins = 0;; ++spins) {
                bool all = true;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (need[u] && (v[u] >> 32) != epoch) {
                        v[u] = ld_u64(p.u + kb + u * NJ_THREADS);
                        all = all && (v[u] >> 32) == epoch;
                    }
                if (all) break;
                if (spins > NJ_SPIN_LIMIT) {
                    S.misc[0] = 1;
                    __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, NJ_AGENT);
                    break;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = kb + u * NJ_THREADS;
                if (k >= n_cur) continue;
                float nd = 0.0f; // the addend of a cluster that takes no part: +0, the identity of the chain
                if (need[u]) {
                    const float uu = __uint_as_float((uint32_t)v[u]);
                    nd = __fmul_rn(__fsub_rn(uu, Dij), 0.5f);
                    S.sum[k] = __fadd_rn(__fsub_rn(S.sum[k], uu), nd); // ck.sum -= Dik + Djk; ck.sum += the new Dik
                }
                S.tmp[k] = nd;
            }
        }
        for (int t = n_cur + tid; t < ordered_sum_padded(n_cur); t += NJ_THREADS) S.tmp[t] = 0.0f;
        if (tid == 0) {
            S.sum[mi] = nan; // until the chain has added it up
            if (mj != mi) { S.act[mj] = 0; S.sum[mj] = nan; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's write-through stores have landed before the next candidate goes out
        __syncthreads();
        if (S.misc[0]) return;
        chain_pending = true;
        --n_act;
        NJ_LAP(4)
    }

    // join the two clusters that remain (NeighborJoining.cpp:112)
    if (w == 0 && tid == 0) {
        int first = -1, second = -1;
        for (int k = 0; k < n_cur; ++k)
            if (S.act[k]) {
                if (first < 0) first = k;
                else if (second < 0) second = k;
            }
        a.left[iter] = a.node[first];
        a.right[iter] = second >= 0 ? a.node[second] : a.node[first];
        if (p.prof)
            for (int x = 0; x < 6; ++x) p.prof[x] = lap[x];
    }
#undef NJ_LAP
}

// The clusters' first sums (NeighborJoining.cpp:44-55: the distances of row i added up in ascending j, j != i): a
// workgroup per row gathers the row into LDS, one wave adds it up (ordered_sum.h).
__global__ __launch_bounds__(256) void nj_init_rows_kernel(NjArgs a, int pad)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* row = (float*)smem;
    const int i = blockIdx.x, n = a.n, tid = threadIdx.x;
    for (int t = tid; t < pad; t += 256) row[t] = t < n - 1 ? a.D[tri(i, t < i ? t : t + 1)] : 0.0f;
    __syncthreads();
    if (tid < 64) {
        const float s = wave_ordered_sum(row, n - 1);
        if (tid == 0) {
            a.sum[i] = s;
            a.node[i] = i;
            a.active[i] = 1;
        }
    }
}

int nj_loop_cap(int n) { return ordered_sum_padded(n); }

size_t nj_loop_lds_bytes(int cap) { return (size_t)cap * 9 + 20 * sizeof(Cand) + (4 + NJ_THREADS + 1) * sizeof(int); }

// Workgroups of the launch: one per CU (all resident at once); 0 = the kernel does not fit a CU
hipError_t nj_loop_grid(int cap, int* grid)
{
    *grid = 0;
    int dev = 0, cus = 0, per_cu = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) != hipSuccess) return e;
    const size_t lds = nj_loop_lds_bytes(cap);
    if ((e = hipFuncSetAttribute((const void*)nj_loop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess)
        return e;
    if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, nj_loop_kernel, NJ_THREADS, lds)) != hipSuccess) return e;
    if (per_cu < 1) return hipSuccess;
    *grid = cus;
    return hipSuccess;
}

hipError_t launch_nj_loop(const NjLoopArgs& p, int grid, hipStream_t stream)
{
    const int pad = ordered_sum_padded(p.a.n - 1);
    hipError_t e = hipFuncSetAttribute((const void*)nj_init_rows_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, pad * 4);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(nj_init_rows_kernel, dim3(p.a.n), dim3(256), (size_t)pad * 4, stream, p.a, pad);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    // A plain launch: with one workgroup per CU (the LDS request leaves room for no second one) the grid is resident as a
    // whole once the CUs are free, which is all hipLaunchCooperativeKernel would check -- and its first use in a process
    // costs ~9 ms (a queue of its own is created).
    hipLaunchKernelGGL(nj_loop_kernel, dim3(grid), dim3(NJ_THREADS), nj_loop_lds_bytes(p.cap), stream, p);
    return hipGetLastError();
}

} // namespace lcsgpu
