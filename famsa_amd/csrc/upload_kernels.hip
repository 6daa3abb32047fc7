// upload_kernels.hip -- builds the device-resident form of an uploaded sequence set on the device:
// the caller's packed codes (1 B/residue, lcsgpu_upload layout) are copied to HBM once and turned
// there into the position-major 64-sequence tiles the LCS kernels stream (byte = code * 8, 16-byte
// chunks, padding = code 22), together with the per-sequence orientation flags (SURVEY note Q).
// On the host this transposition was a byte-wise scatter over the whole set (0.3 s per 10^6
// sequences); here it is one pass at HBM speed.  Sequence k of the set is record order[k] of the
// caller's buffer (order == NULL: record k), so a caller that sorts and drops records never packs
// them on the host (lcsgpu_upload_ordered).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "lcs_kernels.h"

namespace lcsgpu {

// one workgroup per tile; lane = sequence of the tile, the 4 waves stride over the 16-residue chunks
__global__ __launch_bounds__(256) void tiles_fill_kernel(const uint8_t* __restrict__ codes,
                                                         const uint64_t* __restrict__ offsets,
                                                         const int32_t* __restrict__ order,
                                                         const uint64_t* __restrict__ tile_base, int32_t n,
                                                         uint8_t* __restrict__ tiles, int32_t* __restrict__ flags)
{
    const int tile = blockIdx.x, s = threadIdx.x & 63, stripe = threadIdx.x >> 6;
    const int64_t seq = (int64_t)tile * 64 + s;
    const int64_t rec = seq < n ? (order ? order[seq] : seq) : 0;
    const uint64_t off = seq < n ? offsets[rec] : 0;
    const uint32_t len = seq < n ? (uint32_t)(offsets[rec + 1] - off) : 0u;
    const uint64_t base = tile_base[tile];
    const int chunks = (int)((tile_base[tile + 1] - base) >> 10);
    bool bad = false;
    for (int c = stripe; c < chunks; c += 4) {
        uint32_t w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t word = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const uint32_t p = (uint32_t)c * 16 + q * 4 + b;
                uint32_t v = 22;
                if (p < len) {
                    v = codes[off + p];
                    bad |= v >= 32;
                }
                word |= ((v * 8) & 0xFFu) << (8 * b);
            }
            w[q] = word;
        }
        *reinterpret_cast<uint4*>(tiles + base + (uint64_t)c * 1024 + (uint64_t)s * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    if (bad) atomicOr(flags, 1);
}

// A ref is orientation-sensitive iff some 64-bit word w >= 1 that lies fully inside the sequence
// holds 64 copies of one valid residue (only then can tB == ~0 meet a carry-in).
__global__ __launch_bounds__(256) void quirk_flags_kernel(const uint8_t* __restrict__ codes,
                                                          const uint64_t* __restrict__ offsets,
                                                          const int32_t* __restrict__ order, int32_t n,
                                                          uint8_t* __restrict__ quirk)
{
    const int64_t seq = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (seq >= n) return;
    const int64_t rec = order ? order[seq] : seq;
    const uint64_t off = offsets[rec];
    const uint32_t len = (uint32_t)(offsets[rec + 1] - off);
    bool q = false;
    for (uint32_t w = 1; (w + 1) * 64 <= len && !q; ++w) {
        const uint8_t* p = codes + off + (uint64_t)w * 64;
        const uint8_t c = p[0];
        if (c >= 20) continue;
        bool all = true;
        for (int i = 1; i < 64 && all; ++i) all = p[i] == c;
        q = all;
    }
    quirk[seq] = q ? 1 : 0;
}

// The occurrence masks of every sequence (CSequence::ComputeBitMasks, reference core/sequence.cpp:190-201: bits
// only for codes < 20), once per upload: one wave per 64-residue word, 20 ballots, lane c keeps code c's mask.
// The LCS kernels copy the rows of their refs into LDS instead of rebuilding them per workgroup (that was 3 % of
// the hot kernel's instructions at 400 aa and 11 % at 100 aa).  4 bytes per residue of HBM.
__global__ __launch_bounds__(256) void masks_fill_kernel(const uint8_t* __restrict__ codes,
                                                         const uint64_t* __restrict__ offsets,
                                                         const int32_t* __restrict__ order,
                                                         const uint64_t* __restrict__ mask_base, int32_t n,
                                                         uint64_t* __restrict__ masks)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t seq = blockIdx.x;
    const int64_t rec = order ? order[seq] : seq;
    const uint64_t off = offsets[rec];
    const uint32_t len = (uint32_t)(offsets[rec + 1] - off);
    const uint64_t row0 = mask_base[seq];
    const int words = (int)(mask_base[seq + 1] - row0);
    for (int w = wave; w < words; w += 4) {
        const uint32_t p = (uint32_t)w * 64 + lane;
        const uint32_t code = p < len ? codes[off + p] : 0xFFu;
        uint64_t mine = 0;
#pragma unroll
        for (int c = 0; c < 20; ++c) {
            const uint64_t b = __ballot(code == (uint32_t)c);
            if (lane == c) mine = b;
        }
        if (lane < 32) masks[(row0 + w) * 32 + lane] = mine;
    }
}

hipError_t launch_build_set(const uint8_t* codes, const uint64_t* offsets, const int32_t* order, const uint64_t* tile_base, int32_t n,
                            uint8_t* tiles, uint8_t* quirk, int32_t* flags, const uint64_t* mask_base, uint64_t* masks,
                            hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    const int n_tiles = (n + 63) / 64;
    hipLaunchKernelGGL(tiles_fill_kernel, dim3(n_tiles), dim3(256), 0, stream, codes, offsets, order, tile_base, n, tiles, flags);
    hipLaunchKernelGGL(quirk_flags_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, codes, offsets, order, n, quirk);
    hipLaunchKernelGGL(masks_fill_kernel, dim3(n), dim3(256), 0, stream, codes, offsets, order, mask_base, n, masks);
    return hipGetLastError();
}

} // namespace lcsgpu
