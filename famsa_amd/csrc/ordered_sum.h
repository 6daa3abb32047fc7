// ordered_sum.h -- the float sum  s = (((0 + x0) + x1) + x2) + ...  rounded after every addition, computed by one wave
// faster than one addition after the other, and bit for bit the same.  Device code only.
//
// NeighborJoining::computeTree (reference tree/NeighborJoining.cpp:44-55, 88-108) adds distances up in ascending
// cluster order, in float: n dependent additions per merge, 8-13 cycles each on one lane of a CDNA4 SIMD -- 58 of the
// 111 ms of hemopexin's resident NJ launch when done that way (profiles/nj_r06.txt).  The additions are not
// associative, but most of them happen while the sum stays inside one binade [2^e, 2^(e+1)): there the sum is an
// integer S (24 bits) times u = 2^(e-23), and
//
//      fl(S u + x) = (S + round(x / u)) u        while the exact sum stays inside [2^e, 2^(e+1)),
//
// round() to nearest -- a tie (x / u = k + 1/2 exactly) goes to the neighbour that makes S + k even, the only place
// where the sum so far decides.  x / u is exact (a power of two), so are floor and the fraction.  So for a block of
// addends without a tie, whose steps up added to S stay below 2^24 and whose steps down leave S above 2^23 (every
// partial sum, in ANY order, lies between the two; one step of margin at the lower edge when an addend is negative:
// below 2^e the grid is u / 2 and the exact sum may lie half a step under the rounded one), the sequential result is S
// plus the integer total of the round(x / u): 64 lanes, four addends each, two wave reductions.  A block of 256 that
// has a tie, a huge or non-finite addend, or that could leave the binade is taken in pieces of 64 and, failing that
// too, one addition after the other: nothing is approximated anywhere (tests/gpu_src/ordered_sum_check.hip compares
// bit patterns with a host loop over vectors made to hit every branch).
//
// What it buys (profiles/nj_r06.txt): a block costs ~650 cycles of latency whatever its length and a plain addition 8,
// a tie turns up every ~2^11 addends of NJ's kind (ten to twelve fraction bits below u) and the binade changes
// log2(n) times per sum -- 39 ms instead of 58 inside the loaded NJ launch, no gain on an idle CU.  Variants that were
// built, found exact and slower there: prefix sums over 512 addends with the offending addend added for real and
// the rest continued (a round costs 1400+ cycles, and every tie and change of binade costs a round), block lengths
// guessed from the distance to the next power of two (ties cut the long blocks down).
//
// Call with all 64 lanes of ONE wave; `x` (LDS or global) must be readable and +0.0f from n up to
// ordered_sum_padded(n).  Every lane returns the sum.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lcsgpu {

__host__ __device__ constexpr int ordered_sum_padded(int n) { return 64 + ((n > 64 ? n - 64 : 0) + 255) / 256 * 256; }

namespace ordered_sum_detail {

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v)
{
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xF, false); // rows not written add 0
}

__device__ __forceinline__ uint32_t wave_total(uint32_t v) // the wave's sum, wave-uniform
{
    v = dpp_add<0xB1, 0xF>(v);  // quad_perm 1,0,3,2
    v = dpp_add<0x4E, 0xF>(v);  // quad_perm 2,3,0,1
    v = dpp_add<0x141, 0xF>(v); // row_half_mirror
    v = dpp_add<0x140, 0xF>(v); // row_mirror: every lane of a row holds the row's total
    v = dpp_add<0x142, 0xA>(v); // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xC>(v); // row_bcast:31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// s += x[t .. t + 64 U) if the block qualifies (see the header); false: s is unchanged
template <int U>
__device__ __forceinline__ bool block(const float* x, int t, float& s)
{
    const uint32_t sb = __float_as_uint(s);
    const uint32_t eb = (sb >> 23) & 0xff;                         // biased exponent
    if ((sb >> 31) || eb < 27 || eb > 227) return false;           // negative, zero, denormal, inf, NaN or extreme: plain additions
    const uint32_t S = (sb & 0x7fffffu) | 0x800000u;
    const float scale = __uint_as_float((277u - eb) << 23);        // 1 / u = 2^(23 - e)
    const float lim = __uint_as_float((eb + 1u) << 23);            // 2^(e + 1)
    const int lane = threadIdx.x & 63;
    uint32_t up = 0, down = 0; // the steps up and the steps down, added up apart: every partial sum lies between S - down and S + up
    bool bad = false, neg = false;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const float v = x[t + lane + 64 * u];
        bad = bad || !(__builtin_fabsf(v) < lim);                  // too large, inf, NaN
        neg = neg || v < 0.0f;
        const float y = v * scale;                                 // exact
        const float fl = __builtin_floorf(y);
        const float fr = y - fl;                                   // exact, in [0, 1)
        bad = bad || fr == 0.5f;                                   // a tie: the order decides
        const int r = (int)fl + (fr > 0.5f ? 1 : 0);
        if (r >= 0) up += (uint32_t)r;
        else down += (uint32_t)(-r);
    }
    if (__ballot(bad)) return false;
    const uint64_t hi = (uint64_t)S + wave_total(up);
    if (hi >= (1u << 24)) return false;                            // a partial sum could leave the binade upwards
    uint32_t dn = 0;
    if (__ballot(neg)) {
        // ... or downwards: one step of margin, because below 2^e the grid is u / 2 and the exact sum of a negative
        // addend may lie half a step under the rounded one (also when the addend itself rounds to no step at all)
        dn = wave_total(down);
        if ((uint64_t)dn + (1u << 23) + 1 > (uint64_t)S) return false;
    }
    s = __uint_as_float((eb << 23) | ((uint32_t)(hi - dn) & 0x7fffffu));
    return true;
}

} // namespace ordered_sum_detail

// stats (optional, 3 counters of the calling wave): blocks of 256 taken at once, pieces of 64 taken at once, pieces added up one by one
__device__ __forceinline__ float wave_ordered_sum(const float* x, int n, uint32_t* stats = nullptr)
{
    using namespace ordered_sum_detail;
    float s = 0.0f;
    int t = 0;
    auto plain64 = [&] { // 64 additions one after the other, 8 loads in flight (the +0.0f beyond n change nothing: s is never -0.0f)
        for (int k = 0; k < 64; k += 8) {
            float g[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) g[q] = x[t + k + q];
#pragma unroll
            for (int q = 0; q < 8; ++q) s = __fadd_rn(s, g[q]);
        }
    };
    plain64(); // a young sum changes its binade every few additions
    t = 64;
    while (t < n) {
        if (block<4>(x, t, s)) {
            t += 256;
            if (stats) ++stats[0];
            continue;
        }
        for (int c = 0; c < 4; ++c, t += 64) {
            if (t >= n) continue;
            if (block<1>(x, t, s)) {
                if (stats) ++stats[1];
                continue;
            }
            if (stats) ++stats[2];
            plain64();
        }
    }
    return s;
}

} // namespace lcsgpu
