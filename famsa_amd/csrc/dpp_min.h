// dpp_min.h -- wave-level (value, index) first-minimum reductions by DPP, shared by the tree reducers
// (tree_kernels.hip, upgma_batch_kernels.hip).  Device code only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lcsgpu {

// ---- wave-level (value, index) minimum by DPP ------------------------------------------------------------------------
// __shfl_xor compiles to ds_bpermute_b32 -- a trip through the LDS crossbar per value and step -- and a reduction of a
// (float, index) pair took ~0.6 us per wave; the merge kernels do four of them per merge.  The data-parallel-primitive
// controls of gfx9 move a value between lanes inside the VALU (quad_perm, row_half_mirror, row_mirror: a butterfly inside
// a row of 16 lanes; row_bcast:15 / row_bcast:31: the rows' totals into the next row / the upper half), six steps, the
// result in lane 63, read out as a wave-uniform value.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xF, false); // lanes not written keep v
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void dpp_min_step(float& d, uint32_t& j)
{
    const float d2 = __uint_as_float(dpp_u32<CTRL, ROW_MASK>(__float_as_uint(d)));
    const uint32_t j2 = dpp_u32<CTRL, ROW_MASK>(j);
    if (d2 < d || (d2 == d && j2 < j)) { d = d2; j = j2; }
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void dpp_min_step3(float& d, uint32_t& j, uint32_t& nr)
{
    const float d2 = __uint_as_float(dpp_u32<CTRL, ROW_MASK>(__float_as_uint(d)));
    const uint32_t j2 = dpp_u32<CTRL, ROW_MASK>(j), n2 = dpp_u32<CTRL, ROW_MASK>(nr);
    if (d2 < d || (d2 == d && j2 < j)) { d = d2; j = j2; nr = n2; }
}
constexpr int DPP_QUAD_1032 = 0xB1, DPP_QUAD_2301 = 0x4E, DPP_ROW_HALF_MIRROR = 0x141, DPP_ROW_MIRROR = 0x140,
              DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143;
// the wave's (smaller value, then smaller index): in every lane of the last row, returned wave-uniform
__device__ __forceinline__ void wave_first_min(float& d, uint32_t& j)
{
    dpp_min_step<DPP_QUAD_1032, 0xF>(d, j);
    dpp_min_step<DPP_QUAD_2301, 0xF>(d, j);
    dpp_min_step<DPP_ROW_HALF_MIRROR, 0xF>(d, j);
    dpp_min_step<DPP_ROW_MIRROR, 0xF>(d, j);
    dpp_min_step<DPP_ROW_BCAST15, 0xA>(d, j);
    dpp_min_step<DPP_ROW_BCAST31, 0xC>(d, j);
    d = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(d), 63));
    j = (uint32_t)__builtin_amdgcn_readlane((int)j, 63);
}
__device__ __forceinline__ void wave_first_min3(float& d, uint32_t& j, uint32_t& nr)
{
    dpp_min_step3<DPP_QUAD_1032, 0xF>(d, j, nr);
    dpp_min_step3<DPP_QUAD_2301, 0xF>(d, j, nr);
    dpp_min_step3<DPP_ROW_HALF_MIRROR, 0xF>(d, j, nr);
    dpp_min_step3<DPP_ROW_MIRROR, 0xF>(d, j, nr);
    dpp_min_step3<DPP_ROW_BCAST15, 0xA>(d, j, nr);
    dpp_min_step3<DPP_ROW_BCAST31, 0xC>(d, j, nr);
    d = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(d), 63));
    j = (uint32_t)__builtin_amdgcn_readlane((int)j, 63);
    nr = (uint32_t)__builtin_amdgcn_readlane((int)nr, 63);
}
// the same over the first 16 lanes only (one row): quad butterflies + the two mirrors; result from lane 0
__device__ __forceinline__ void row_first_min3(float& d, uint32_t& j, uint32_t& nr)
{
    dpp_min_step3<DPP_QUAD_1032, 0xF>(d, j, nr);
    dpp_min_step3<DPP_QUAD_2301, 0xF>(d, j, nr);
    dpp_min_step3<DPP_ROW_HALF_MIRROR, 0xF>(d, j, nr);
    dpp_min_step3<DPP_ROW_MIRROR, 0xF>(d, j, nr);
    d = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(d)));
    j = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);
    nr = (uint32_t)__builtin_amdgcn_readfirstlane((int)nr);
}

} // namespace lcsgpu
