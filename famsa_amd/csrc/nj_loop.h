// nj_loop.h -- neighbour joining as one resident launch: nj_loop_kernels.hip
#pragma once
#include "lcs_kernels.h"

namespace lcsgpu {

struct NjLoopArgs {
    NjArgs a;          // a.D: the triangle, 16-byte aligned, padded to a multiple of 4 floats
    float* D2;         // a second triangle of the same size (the rows that are gone are squeezed out from time to time)
    uint64_t* slots;   // [2 x 2 x grid] the workgroups' posted candidates (two buffers, two tagged words each), zeroed
    uint64_t* u;       // [n] Dik + Djk of the current merge, tagged, zeroed
    int32_t* err;      // set when a word did not arrive in time, zeroed
    long long* prof;   // [6] or NULL: workgroup 0's time in squeezes / scan + chain / exchange / updates / polls (10 ns ticks)
    int32_t cap;       // ordered_sum_padded(n): rows the LDS copies are laid out for (n <= NJ_LOOP_MAX_N)
    int32_t compact_min; // no squeezing below this many rows
};
constexpr int NJ_LOOP_MAX_N = 16384;
int nj_loop_cap(int n);
hipError_t nj_loop_grid(int cap, int* grid); // workgroups of the launch (one per CU); 0: the kernel does not fit
hipError_t launch_nj_loop(const NjLoopArgs& p, int grid, hipStream_t stream); // includes the initial sums

} // namespace lcsgpu
