// ubench.hip -- instruction-throughput ceilings for the LCS word-step on gfx950 (dev tool).
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -Xclang -target-feature -Xclang -load-store-opt scripts/ubench.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ uint32_t andn(uint32_t v, uint32_t n) { return __builtin_amdgcn_bitop3_b32(v, n, 0u, 0x30); }
__device__ __forceinline__ uint32_t or_and(uint32_t s, uint32_t v, uint32_t n) { return __builtin_amdgcn_bitop3_b32(s, v, n, 0xF8); }

// A: VALU only -- NCH independent carry chains of W words, masks in registers
template <int NCH, int W>
__global__ __launch_bounds__(256) void k_valu(uint32_t* out, int iters, uint32_t seed)
{
    uint32_t X[NCH][2 * W];
    uint32_t m0 = seed * (threadIdx.x + 1), m1 = ~m0 * 2654435761u;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 2 * W; ++j) X[c][j] = ~0u - c - j;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            unsigned cin = 0, co;
#pragma unroll
            for (int j = 0; j < W; ++j) {
                uint32_t V = X[c][2 * j];
                uint32_t s = __builtin_addc(V, andn(V, m0), cin, &co);
                X[c][2 * j] = or_and(s, V, m0); cin = co;
                V = X[c][2 * j + 1];
                s = __builtin_addc(V, andn(V, m1), cin, &co);
                X[c][2 * j + 1] = or_and(s, V, m1); cin = co;
            }
        }
        m0 = m0 * 1664525u + 1013904223u; m1 ^= m0;   // 2 extra VALU per iter
    }
    uint32_t r = 0;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 2 * W; ++j) r += X[c][j];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

// B: plain independent v_and / v_bitop3 / v_add streams
template <int KIND>
__global__ __launch_bounds__(256) void k_simple(uint32_t* out, int iters, uint32_t seed)
{
    uint32_t a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = seed + i * threadIdx.x;
    uint32_t m = seed ^ threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) a[i] = (a[i] & m) ^ 0x5a5a5a5au;            // may fuse; see asm
                if (KIND == 1) a[i] = __builtin_amdgcn_bitop3_b32(a[i], m, a[(i + 1) & 15], 0x96);
                if (KIND == 2) a[i] = a[i] + m;
                if (KIND == 3) { unsigned co; a[i] = __builtin_addc(a[i], m, (unsigned)(a[(i + 1) & 15] >> 31), &co); }
            }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

// C: LDS gather only: ds_read_b64 at (code*8 + row*256), codes random per lane
typedef __attribute__((address_space(3))) unsigned char lds_u8;
typedef __attribute__((address_space(3))) uint64_t lds_u64;
template <int ROWS>
__global__ __launch_bounds__(256) void k_lds(uint32_t* out, int iters, uint32_t seed)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    for (int i = threadIdx.x; i < ROWS * 32; i += 256) ((uint64_t*)smem)[i] = i * 0x9E3779B97F4A7C15ull;
    __syncthreads();
    uint32_t h = seed * (threadIdx.x * 2 + 1);
    uint64_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        h = h * 1664525u + 1013904223u;
        const uint32_t code8 = ((h >> 16) % 20u) * 8u;
        const lds_u8* row = (const lds_u8*)smem + code8;
#pragma unroll
        for (int j = 0; j < ROWS; ++j) acc ^= *(const lds_u64*)(row + j * 256);
    }
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)acc ^ (uint32_t)(acc >> 32);
}

template <typename F>
static double time_ms(F launch, int reps = 5)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    double best = 1e30;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const double ghz = p.clockRate / 1e6;
    const int cus = p.multiProcessorCount;
    printf("%s CUs %d clock %.2f GHz\n", p.gcnArchName, cus, ghz);
    uint32_t* out; CK(hipMalloc(&out, (size_t)cus * 64 * 256 * 4));
    for (int wpc : {4, 8, 16}) { // waves per CU = blocks/CU * 4
        const int blocks = cus * (wpc / 4);
        printf("-- %d waves/CU (%d/SIMD)\n", wpc, wpc / 4);
        {
            const int iters = 2000; const double instr = (double)iters * (28 * 6 + 2);
            double ms = time_ms([&] { hipLaunchKernelGGL((k_valu<4, 7>), dim3(blocks), dim3(256), 0, 0, out, iters, 12345u); });
            double waves_per_simd = wpc / 4.0;
            printf("valu wordstep<4,7>: %.3f ms  -> %.2f cycles/VALU-instr/SIMD (at %.2f GHz), %.1f Tword-step/s chip\n", ms,
                   ms * 1e-3 * ghz * 1e9 / (instr * waves_per_simd), ghz, (double)blocks * 256 * iters * 28 / (ms * 1e-3) / 1e12);
        }
        const char* names[4] = {"and+xor(2 ops)", "bitop3", "add_u32", "addc"};
        for (int kind = 0; kind < 4; ++kind) {
            const int iters = 1000; double instr = (double)iters * 128 * (kind == 0 ? 2 : 1);
            double ms = 0;
            if (kind == 0) ms = time_ms([&] { hipLaunchKernelGGL((k_simple<0>), dim3(blocks), dim3(256), 0, 0, out, iters, 77u); });
            if (kind == 1) ms = time_ms([&] { hipLaunchKernelGGL((k_simple<1>), dim3(blocks), dim3(256), 0, 0, out, iters, 77u); });
            if (kind == 2) ms = time_ms([&] { hipLaunchKernelGGL((k_simple<2>), dim3(blocks), dim3(256), 0, 0, out, iters, 77u); });
            if (kind == 3) ms = time_ms([&] { hipLaunchKernelGGL((k_simple<3>), dim3(blocks), dim3(256), 0, 0, out, iters, 77u); });
            printf("%-16s: %.3f ms -> %.2f cycles/instr/SIMD\n", names[kind], ms, ms * 1e-3 * ghz * 1e9 / (instr * (wpc / 4.0)));
        }
        {
            const int iters = 20000; const double reads = (double)iters * 28;
            double ms = time_ms([&] { hipLaunchKernelGGL((k_lds<28>), dim3(blocks), dim3(256), 28 * 256, 0, out, iters, 999u); });
            printf("lds gather b64    : %.3f ms -> %.2f cycles per wave-read per CU, %.1f Tread-lanes/s chip\n", ms,
                   ms * 1e-3 * ghz * 1e9 / (reads * wpc), (double)blocks * 256 * reads / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
