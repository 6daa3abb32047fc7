// lcs_kernels.h -- internal interface between the C-ABI layer (lcsgpu_api.hip) and the
// gfx950 kernels (lcs_kernels.hip).  Not installed; the public boundary is include/lcsgpu.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lcsgpu {

enum { MODE_RECT = 0, MODE_TRIANGLE = 1 };

struct RowsArgs {
    // the uploaded sequence set (device)
    const uint8_t* tiles;      // position-major residue store, bytes = code*8
    const uint64_t* tile_base; // byte offset of each 64-sequence tile
    const uint32_t* lens;      // length per sequence
    // refs (bit-mask side): ids ref_ids[k] or ref_begin + k, k < n_refs
    const int32_t* ref_ids;
    const int64_t* ref_rows; // RECT: output row of ref k (else row0 + k)
    int32_t ref_begin;
    int32_t n_refs;
    // partners (streamed side): ids col_ids[c] or col_begin + c, c < n_cols
    const int32_t* col_ids;
    int32_t col_begin;
    int32_t n_cols;
    // output
    void* out;
    int64_t ld;         // RECT: out[row*ld + c]
    int64_t row0;       // RECT: first output row for contiguous refs
    int64_t out_offset; // TRIANGLE: out[rid*(rid-1)/2 + c - out_offset]
    int32_t elem_size;  // 2 or 4
    int32_t mode;
    int32_t refs_per_block;
};

// instantiated word counts: exact 1..16, even 18..32; 0 = needs the long-sequence path
int bv_class(uint32_t len);
int quirk_bv_class(uint32_t len);
int refs_per_block(int bv, bool quirk);
hipError_t launch_rows(int bv, bool quirk, const RowsArgs& a, int grid_x, int grid_y, hipStream_t stream);


struct RowMin {
    double dist;
    int64_t index;
};

// per-row minima over a triangle slice (see lcsgpu_row_minima_dev in include/lcsgpu.h)
hipError_t launch_row_minima(const void* tri, int elem_size, int32_t row_begin, int32_t row_end,
                             const uint32_t* lens, const double* pow_table, int kind, RowMin* out,
                             hipStream_t stream);

} // namespace lcsgpu
