// fasttree_kernels.h -- internal interface between lcsgpu_fasttree.hip and the kernels behind the batched calls of the
// FastTree recursion (tree_kernels.hip, clarans_kernels.hip): all splits of a level in one launch wave.  Not installed.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lcs_kernels.h"

namespace lcsgpu {

// ---- seed assignment of several evaluations in one launch (tree_kernels.hip) ----
struct AssignPiece {
    int64_t out0;    // element offset of the piece's LCS rectangle (n_seeds rows of n_cols)
    int32_t col0;    // first column: position in the launch's concatenated column list
    int32_t n_cols;
    int32_t seed0;   // first seed: position in the concatenated seed list
    int32_t n_seeds;
};
hipError_t launch_assign_seeds_batch(const void* lcs, int elem_size, const AssignPiece* pieces, int32_t n_pieces, const int32_t* seed_ids,
                                     const int32_t* col_ids, int64_t n_cols, const uint32_t* lens, const float* pow_f32, int kind,
                                     float* dist, int32_t* assign, hipStream_t stream);

// ---- CLARANS for many samples at once: one workgroup runs a sample's whole chain of local searches (clarans_kernels.hip) ----
constexpr int CLARANS_STATE_WORDS = 32; // a chain's state block (ST_* in clarans_kernels.hip)
struct ClaransChain {
    ClaransArgs a;        // the search's buffers; a.draws / a.draws_len: the shape's pre-drawn step positions
    const int32_t* perm;  // [num_local][n_elems] before local search t, position i takes what position perm[t][i] held
    const int32_t* ids;   // [n_elems] the sample's sequence ids
    int32_t* best;        // [n_medoids] the medoids of the cheapest search so far
    int64_t tri0;         // element offset of the sample's packed LCS triangle in the launch's triangle buffer
    int32_t num_local;
};
hipError_t launch_subset_distances_batch(const void* lcs, int elem_size, const ClaransChain* chains, int n_chains, int max_n,
                                         const uint32_t* lens, const float* pow_f32, int kind, hipStream_t stream);
// every chain in its own workgroup, to its end -- or until its pre-drawn positions run out, or for slice_us microseconds (0 = no limit)
hipError_t launch_clarans_chains(const ClaransChain* chains, int n_chains, int max_medoids, int slice_us, hipStream_t stream);

} // namespace lcsgpu
