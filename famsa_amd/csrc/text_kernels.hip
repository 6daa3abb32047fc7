// text_kernels.hip -- the rows of -dist_export as TEXT, made on the device (gfx950).
//
// What the reference does per row on the host (tree/DistanceCalculator.cpp:88-113): "<id>," then every value through
// NumericConversions::Double2PChar(v, 6) (utils/conversion.h:109-119) separated by ',', the last separator replaced by
// '\n'.  Values are floats: (float) Transform<double, kind>(lcs, len_i, len_j) resp. Transform<float, pairwise_identity>
// (DistanceCalculator.cpp:36-76, tree/AbstractTreeGenerator.hpp:28-82).  Here a row block's LCS rectangle never leaves
// HBM as numbers: three small kernels turn it into the block's final bytes --
//   text_len_kernel   one workgroup per (row, segment of 1024 values): the segment's text length
//   text_scan_*       within-row prefix (one wave per row), then the rows' start offsets (one workgroup)
//   text_write_kernel the same workgroups again: every thread formats its 4 values into LDS at the scanned offsets,
//                     the workgroup copies the segment to its place with aligned 16-byte stores
// Bound: nothing on the chip (9 B of text per 2 B read; a block's kernels take well under a millisecond per 100 MB);
// the stage is bound by the host's write of the text.  All arithmetic that decides a byte is the reference's, in the
// reference's types and order: double division on the host-built pow table, round to float, back to double,
// a = (int64) v, b = (int64) ((1.0 + (v - a)) * 1e6 + 0.5) without contraction (-ffp-contract=off), the conversion of
// an out-of-range double as x86-64's cvttsd2si does it (INT64_MIN), so the lcs == 0 value prints as
// 9223372036854775808.223372036854775808 here too.
#include "lcs_kernels.h"

namespace lcsgpu {

namespace {

constexpr int TEXT_THREADS = 256;
constexpr int TEXT_PER_THREAD = 4;
constexpr int TEXT_SEG = TEXT_THREADS * TEXT_PER_THREAD; // values per workgroup
constexpr int TEXT_MAX_VALUE = 39; // 19 + '.' + 18 digits + separator: the saturated value
constexpr int TEXT_LDS = TEXT_SEG * TEXT_MAX_VALUE + 32;

// (int64) v as the reference's x86-64 build converts (cvttsd2si): out of range or NaN -> INT64_MIN
__device__ __forceinline__ long long trunc_i64(double v)
{
    if (!(v >= -9223372036854775808.0 && v < 9223372036854775808.0)) return (long long)0x8000000000000000ull;
    return (long long)v;
}

__device__ __forceinline__ int n_digits(unsigned long long v)
{
    if (v <= 0xFFFFFFFFull) {
        uint32_t x = (uint32_t)v;
        int n = 1;
        while (x >= 10u) {
            x /= 10u;
            ++n;
        }
        return n;
    }
    int n = 1;
    while (v >= 10ull) {
        v /= 10ull;
        ++n;
    }
    return n;
}

// the decimal digits of v, the last one at p[end - 1]
__device__ __forceinline__ void put_digits(char* p, int end, unsigned long long v)
{
    while (v > 0xFFFFFFFFull) {
        p[--end] = (char)('0' + (int)(v % 10ull));
        v /= 10ull;
    }
    uint32_t x = (uint32_t)v;
    do {
        p[--end] = (char)('0' + (int)(x % 10u));
        x /= 10u;
    } while (x);
}

struct Parts {
    unsigned long long a, b;
    int na, nb;
};

// Double2PChar(v, 6): the two integers it prints and how many digits each has
__device__ __forceinline__ Parts split_value(float f)
{
    const double v = (double)f;
    const long long a = trunc_i64(v);
    const long long b = trunc_i64((1.0 + (v - (double)a)) * 1000000.0 + 0.5);
    Parts p;
    p.a = (unsigned long long)a;
    p.b = (unsigned long long)b;
    p.na = n_digits(p.a);
    p.nb = n_digits(p.b);
    return p;
}

template <typename T>
__device__ __forceinline__ float value_of(const TextArgs& t, const T* __restrict__ lcs_row, int32_t j, uint32_t len_i)
{
    const uint32_t l = lcs_row[t.where ? t.where[j] : j];
    const uint32_t len_j = t.lens[j];
    if (t.kind == 2) return (float)l / (float)(len_i < len_j ? len_i : len_j); // Transform<float, pairwise_identity>
    const uint32_t indel = len_i + len_j - 2u * l;
    double d;
    if (l == 0) d = 1.7976931348623155e308; // nextafter(DBL_MAX, 0), hpp:61,73
    else d = t.kind == 1 ? t.pow_f64[indel] / (double)l : (double)indel / (double)l;
    return (float)d; // the reference stores the row as floats before printing
}

__device__ __forceinline__ int32_t cols_of_row(const TextArgs& t, int32_t i) { return t.square ? t.n : i; }

template <typename T>
__global__ __launch_bounds__(TEXT_THREADS) void text_len_kernel(TextArgs t)
{
    const int seg = blockIdx.x, r = blockIdx.y;
    const int32_t i = t.row_begin + r;
    const int32_t cols = cols_of_row(t, i);
    const int32_t j0 = seg * TEXT_SEG + (int)threadIdx.x * TEXT_PER_THREAD;
    const uint32_t len_i = t.lens[i];
    const T* lcs_row = (const T*)t.lcs + (int64_t)r * t.ld;
    uint32_t mine = 0;
    for (int k = 0; k < TEXT_PER_THREAD; ++k) {
        const int32_t j = j0 + k;
        if (j >= cols) break;
        const Parts p = split_value(value_of<T>(t, lcs_row, j, len_i));
        mine += (uint32_t)(p.na + p.nb + 1); // the '.' takes the place of b's first digit; + the separator
    }
    for (int off = 32; off; off >>= 1) mine += __shfl_down(mine, off, 64);
    __shared__ uint32_t part[TEXT_THREADS / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t sum = 0;
        for (int w = 0; w < TEXT_THREADS / 64; ++w) sum += part[w];
        if (seg == 0) sum += (uint32_t)(t.id_off[i + 1] - t.id_off[i]) + 1u; // "<id>," (or "<id>\n" for an empty row)
        t.seg_len[(int64_t)r * t.segs + seg] = sum;
    }
}

// one wave per row: seg_len -> exclusive prefix inside the row (in place), row_len = the row's bytes
__global__ __launch_bounds__(256) void text_scan_rows_kernel(TextArgs t)
{
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= t.n_rows) return;
    uint32_t* s = t.seg_len + (int64_t)r * t.segs;
    uint32_t run = 0;
    for (int base = 0; base < t.segs; base += 64) {
        const int k = base + lane;
        const uint32_t v = k < t.segs ? s[k] : 0u;
        uint32_t inc = v;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(inc, off, 64);
            if (lane >= off) inc += up;
        }
        if (k < t.segs) s[k] = run + inc - v;
        run += __shfl(inc, 63, 64);
    }
    if (lane == 0) t.row_len[r] = run;
}

// one workgroup: row_len -> row_start (exclusive, 64-bit), row_start[n_rows] = the block's bytes
__global__ __launch_bounds__(1024) void text_scan_block_kernel(TextArgs t)
{
    __shared__ unsigned long long wave_sum[16];
    __shared__ unsigned long long carry;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < t.n_rows; base += 1024) {
        const int r = base + threadIdx.x;
        const unsigned long long v = r < t.n_rows ? (unsigned long long)t.row_len[r] : 0ull;
        unsigned long long inc = v;
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned long long up = __shfl_up(inc, off, 64);
            if (lane >= off) inc += up;
        }
        if (lane == 63) wave_sum[wave] = inc;
        __syncthreads();
        unsigned long long before = carry;
        for (int w = 0; w < wave; ++w) before += wave_sum[w];
        if (r < t.n_rows) t.row_start[r] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) t.row_start[t.n_rows] = carry;
}

template <typename T>
__global__ __launch_bounds__(TEXT_THREADS) void text_write_kernel(TextArgs t)
{
    __shared__ __align__(16) char lds[TEXT_LDS];
    __shared__ uint32_t wave_sum[TEXT_THREADS / 64];
    const int seg = blockIdx.x, r = blockIdx.y;
    const int32_t i = t.row_begin + r;
    const int32_t cols = cols_of_row(t, i);
    if (seg > 0 && (int64_t)seg * TEXT_SEG >= cols) return;
    const uint32_t len_i = t.lens[i];
    const T* lcs_row = (const T*)t.lcs + (int64_t)r * t.ld;
    unsigned long long dst = t.row_start[r] + t.seg_len[(int64_t)r * t.segs + seg];
    if (seg == 0) { // the row's id and its separator, straight to memory
        const unsigned long long b0 = t.id_off[i], id_len = t.id_off[i + 1] - b0;
        for (unsigned long long k = threadIdx.x; k < id_len; k += TEXT_THREADS) t.out[dst + k] = t.ids[b0 + k];
        if (threadIdx.x == 0) t.out[dst + id_len] = cols == 0 ? '\n' : ',';
        dst += id_len + 1;
    }
    const int32_t j0 = seg * TEXT_SEG + (int)threadIdx.x * TEXT_PER_THREAD;
    Parts p[TEXT_PER_THREAD];
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < TEXT_PER_THREAD; ++k) {
        const int32_t j = j0 + k;
        if (j < cols) {
            p[k] = split_value(value_of<T>(t, lcs_row, j, len_i));
            mine += (uint32_t)(p[k].na + p[k].nb + 1);
        } else {
            p[k].na = p[k].nb = 0;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t up = __shfl_up(inc, off, 64);
        if (lane >= off) inc += up;
    }
    if (lane == 63) wave_sum[wave] = inc;
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (int w = 0; w < TEXT_THREADS / 64; ++w) {
        if (w < wave) before += wave_sum[w];
        total += wave_sum[w];
    }
    // LDS byte k + mis <-> output byte dst + k: the copy below then moves whole aligned 16-byte words
    const uint32_t mis = (uint32_t)((unsigned long long)(uintptr_t)(t.out + dst) & 15ull);
    uint32_t at = mis + before + inc - mine;
#pragma unroll
    for (int k = 0; k < TEXT_PER_THREAD; ++k) {
        if (p[k].na == 0) continue;
        put_digits(lds, (int)at + p[k].na, p[k].a);
        put_digits(lds, (int)at + p[k].na + p[k].nb, p[k].b);
        lds[at + p[k].na] = '.'; // over b's leading digit
        at += (uint32_t)(p[k].na + p[k].nb);
        lds[at++] = (j0 + k == cols - 1) ? '\n' : ',';
    }
    __syncthreads();
    char* const base = t.out + dst - mis; // 16-byte aligned
    const uint32_t end = mis + total, words = (end + 15u) / 16u;
    for (uint32_t w = threadIdx.x; w < words; w += TEXT_THREADS) {
        const uint32_t b0 = w * 16u;
        if (b0 >= mis && b0 + 16u <= end) {
            *reinterpret_cast<uint4*>(base + b0) = *reinterpret_cast<const uint4*>(lds + b0);
        } else {
            const uint32_t lo = b0 < mis ? mis : b0, hi = b0 + 16u < end ? b0 + 16u : end;
            for (uint32_t b = lo; b < hi; ++b) base[b] = lds[b];
        }
    }
}

} // namespace

hipError_t launch_text_block(const TextArgs& t, hipStream_t stream)
{
    if (t.n_rows <= 0) return hipSuccess;
    const dim3 grid((unsigned)t.segs, (unsigned)t.n_rows);
    if (t.elem_size == 2) hipLaunchKernelGGL(text_len_kernel<uint16_t>, grid, dim3(TEXT_THREADS), 0, stream, t);
    else hipLaunchKernelGGL(text_len_kernel<uint32_t>, grid, dim3(TEXT_THREADS), 0, stream, t);
    hipLaunchKernelGGL(text_scan_rows_kernel, dim3((unsigned)((t.n_rows + 3) / 4)), dim3(256), 0, stream, t);
    hipLaunchKernelGGL(text_scan_block_kernel, dim3(1), dim3(1024), 0, stream, t);
    if (t.elem_size == 2) hipLaunchKernelGGL(text_write_kernel<uint16_t>, grid, dim3(TEXT_THREADS), 0, stream, t);
    else hipLaunchKernelGGL(text_write_kernel<uint32_t>, grid, dim3(TEXT_THREADS), 0, stream, t);
    return hipGetLastError();
}

int text_segment_values() { return TEXT_SEG; }
int text_max_value_bytes() { return TEXT_MAX_VALUE; }

} // namespace lcsgpu
