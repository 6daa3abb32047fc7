// ubench_banks.hip -- does the VGPR bank of an instruction's sources decide its issue cost on gfx950?  (dev tool)
// hipcc --offload-arch=gfx950 -O3 scripts/ubench_banks.hip -o /tmp/ubb && /tmp/ubb
// Each kernel runs ITER x 64 copies of ONE instruction with fixed physical registers (inline asm), destinations rotating
// over 8 registers so that no instruction depends on a recent one.  bank(vN) is assumed to be N % 4.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP8(s0, s1, s2, s3, s4, s5, s6, s7) s0 s1 s2 s3 s4 s5 s6 s7
#define BODY(I0, I1, I2, I3, I4, I5, I6, I7) \
    REP8(I0, I1, I2, I3, I4, I5, I6, I7) REP8(I0, I1, I2, I3, I4, I5, I6, I7) REP8(I0, I1, I2, I3, I4, I5, I6, I7) REP8(I0, I1, I2, I3, I4, I5, I6, I7) \
    REP8(I0, I1, I2, I3, I4, I5, I6, I7) REP8(I0, I1, I2, I3, I4, I5, I6, I7) REP8(I0, I1, I2, I3, I4, I5, I6, I7) REP8(I0, I1, I2, I3, I4, I5, I6, I7)
#define CLOB "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "vcc"

#define KERNEL(NAME, I0, I1, I2, I3, I4, I5, I6, I7)                                                   \
    __global__ __launch_bounds__(256) void NAME(uint32_t* out, int iters)                              \
    {                                                                                                  \
        asm volatile("v_mov_b32 v4, 1\n v_mov_b32 v5, 2\n v_mov_b32 v6, 3\n v_mov_b32 v7, 4\n"            \
                     "v_mov_b32 v8, 5\n v_mov_b32 v9, 6\n v_mov_b32 v10, 7\n v_mov_b32 v11, 8\n"          \
                     "v_mov_b32 v12, 9\n v_mov_b32 v13, 10\n v_mov_b32 v14, 11\n v_mov_b32 v15, 12\n" ::: CLOB); \
        for (int it = 0; it < iters; ++it) asm volatile(BODY(I0, I1, I2, I3, I4, I5, I6, I7)::: CLOB);  \
        uint32_t r;                                                                                    \
        asm volatile("v_add_u32 %0, v20, v21\n v_add_u32 %0, %0, v22\n v_add_u32 %0, %0, v23" : "=v"(r)::CLOB); \
        out[blockIdx.x * 256 + threadIdx.x] = r;                                                       \
    }

// 3 VGPR sources, all in bank 0
KERNEL(k_b3_same, "v_bitop3_b32 v20, v4, v8, v12 bitop3:0xf8\n", "v_bitop3_b32 v21, v4, v8, v12 bitop3:0xf8\n", "v_bitop3_b32 v22, v4, v8, v12 bitop3:0xf8\n",
       "v_bitop3_b32 v23, v4, v8, v12 bitop3:0xf8\n", "v_bitop3_b32 v24, v4, v8, v12 bitop3:0xf8\n", "v_bitop3_b32 v25, v4, v8, v12 bitop3:0xf8\n",
       "v_bitop3_b32 v26, v4, v8, v12 bitop3:0xf8\n", "v_bitop3_b32 v27, v4, v8, v12 bitop3:0xf8\n")
// 3 VGPR sources in banks 0, 1, 2
KERNEL(k_b3_diff, "v_bitop3_b32 v20, v4, v9, v14 bitop3:0xf8\n", "v_bitop3_b32 v21, v4, v9, v14 bitop3:0xf8\n", "v_bitop3_b32 v22, v4, v9, v14 bitop3:0xf8\n",
       "v_bitop3_b32 v23, v4, v9, v14 bitop3:0xf8\n", "v_bitop3_b32 v24, v4, v9, v14 bitop3:0xf8\n", "v_bitop3_b32 v25, v4, v9, v14 bitop3:0xf8\n",
       "v_bitop3_b32 v26, v4, v9, v14 bitop3:0xf8\n", "v_bitop3_b32 v27, v4, v9, v14 bitop3:0xf8\n")
// two of three in the same bank
KERNEL(k_b3_two, "v_bitop3_b32 v20, v4, v8, v14 bitop3:0xf8\n", "v_bitop3_b32 v21, v4, v8, v14 bitop3:0xf8\n", "v_bitop3_b32 v22, v4, v8, v14 bitop3:0xf8\n",
       "v_bitop3_b32 v23, v4, v8, v14 bitop3:0xf8\n", "v_bitop3_b32 v24, v4, v8, v14 bitop3:0xf8\n", "v_bitop3_b32 v25, v4, v8, v14 bitop3:0xf8\n",
       "v_bitop3_b32 v26, v4, v8, v14 bitop3:0xf8\n", "v_bitop3_b32 v27, v4, v8, v14 bitop3:0xf8\n")
// bitop3 with two VGPR sources + inline constant: same bank / different banks
KERNEL(k_b2c_same, "v_bitop3_b32 v20, v4, v8, 0 bitop3:0x30\n", "v_bitop3_b32 v21, v4, v8, 0 bitop3:0x30\n", "v_bitop3_b32 v22, v4, v8, 0 bitop3:0x30\n",
       "v_bitop3_b32 v23, v4, v8, 0 bitop3:0x30\n", "v_bitop3_b32 v24, v4, v8, 0 bitop3:0x30\n", "v_bitop3_b32 v25, v4, v8, 0 bitop3:0x30\n",
       "v_bitop3_b32 v26, v4, v8, 0 bitop3:0x30\n", "v_bitop3_b32 v27, v4, v8, 0 bitop3:0x30\n")
KERNEL(k_b2c_diff, "v_bitop3_b32 v20, v4, v9, 0 bitop3:0x30\n", "v_bitop3_b32 v21, v4, v9, 0 bitop3:0x30\n", "v_bitop3_b32 v22, v4, v9, 0 bitop3:0x30\n",
       "v_bitop3_b32 v23, v4, v9, 0 bitop3:0x30\n", "v_bitop3_b32 v24, v4, v9, 0 bitop3:0x30\n", "v_bitop3_b32 v25, v4, v9, 0 bitop3:0x30\n",
       "v_bitop3_b32 v26, v4, v9, 0 bitop3:0x30\n", "v_bitop3_b32 v27, v4, v9, 0 bitop3:0x30\n")
// VOP2 and: same bank / different banks
KERNEL(k_and_same, "v_and_b32 v20, v4, v8\n", "v_and_b32 v21, v4, v8\n", "v_and_b32 v22, v4, v8\n", "v_and_b32 v23, v4, v8\n",
       "v_and_b32 v24, v4, v8\n", "v_and_b32 v25, v4, v8\n", "v_and_b32 v26, v4, v8\n", "v_and_b32 v27, v4, v8\n")
KERNEL(k_and_diff, "v_and_b32 v20, v4, v9\n", "v_and_b32 v21, v4, v9\n", "v_and_b32 v22, v4, v9\n", "v_and_b32 v23, v4, v9\n",
       "v_and_b32 v24, v4, v9\n", "v_and_b32 v25, v4, v9\n", "v_and_b32 v26, v4, v9\n", "v_and_b32 v27, v4, v9\n")
// carry chain as in the kernel: add_co / addc_co alternating on independent registers, sources same / different banks
KERNEL(k_addc_same, "v_add_co_u32 v20, vcc, v4, v8\n", "v_and_b32 v21, v5, v9\n", "v_and_b32 v22, v6, v10\n", "v_addc_co_u32 v23, vcc, v4, v8, vcc\n",
       "v_and_b32 v24, v5, v9\n", "v_and_b32 v25, v6, v10\n", "v_addc_co_u32 v26, vcc, v4, v8, vcc\n", "v_and_b32 v27, v5, v9\n")
KERNEL(k_addc_diff, "v_add_co_u32 v20, vcc, v4, v9\n", "v_and_b32 v21, v5, v10\n", "v_and_b32 v22, v6, v11\n", "v_addc_co_u32 v23, vcc, v4, v9, vcc\n",
       "v_and_b32 v24, v5, v10\n", "v_and_b32 v25, v6, v11\n", "v_addc_co_u32 v26, vcc, v4, v9, vcc\n", "v_and_b32 v27, v5, v10\n")

KERNEL(k_and_or, "v_and_or_b32 v20, v4, v9, v14\n", "v_and_or_b32 v21, v4, v9, v14\n", "v_and_or_b32 v22, v4, v9, v14\n", "v_and_or_b32 v23, v4, v9, v14\n", "v_and_or_b32 v24, v4, v9, v14\n", "v_and_or_b32 v25, v4, v9, v14\n", "v_and_or_b32 v26, v4, v9, v14\n", "v_and_or_b32 v27, v4, v9, v14\n")
KERNEL(k_or3, "v_or3_b32 v20, v4, v9, v14\n", "v_or3_b32 v21, v4, v9, v14\n", "v_or3_b32 v22, v4, v9, v14\n", "v_or3_b32 v23, v4, v9, v14\n", "v_or3_b32 v24, v4, v9, v14\n", "v_or3_b32 v25, v4, v9, v14\n", "v_or3_b32 v26, v4, v9, v14\n", "v_or3_b32 v27, v4, v9, v14\n")
KERNEL(k_bfi, "v_bfi_b32 v20, v4, v9, v14\n", "v_bfi_b32 v21, v4, v9, v14\n", "v_bfi_b32 v22, v4, v9, v14\n", "v_bfi_b32 v23, v4, v9, v14\n", "v_bfi_b32 v24, v4, v9, v14\n", "v_bfi_b32 v25, v4, v9, v14\n", "v_bfi_b32 v26, v4, v9, v14\n", "v_bfi_b32 v27, v4, v9, v14\n")
KERNEL(k_xad, "v_xad_u32 v20, v4, v9, v14\n", "v_xad_u32 v21, v4, v9, v14\n", "v_xad_u32 v22, v4, v9, v14\n", "v_xad_u32 v23, v4, v9, v14\n", "v_xad_u32 v24, v4, v9, v14\n", "v_xad_u32 v25, v4, v9, v14\n", "v_xad_u32 v26, v4, v9, v14\n", "v_xad_u32 v27, v4, v9, v14\n")
KERNEL(k_add3, "v_add3_u32 v20, v4, v9, v14\n", "v_add3_u32 v21, v4, v9, v14\n", "v_add3_u32 v22, v4, v9, v14\n", "v_add3_u32 v23, v4, v9, v14\n", "v_add3_u32 v24, v4, v9, v14\n", "v_add3_u32 v25, v4, v9, v14\n", "v_add3_u32 v26, v4, v9, v14\n", "v_add3_u32 v27, v4, v9, v14\n")
KERNEL(k_xor, "v_xor_b32 v20, v4, v9\n", "v_xor_b32 v21, v4, v9\n", "v_xor_b32 v22, v4, v9\n", "v_xor_b32 v23, v4, v9\n", "v_xor_b32 v24, v4, v9\n", "v_xor_b32 v25, v4, v9\n", "v_xor_b32 v26, v4, v9\n", "v_xor_b32 v27, v4, v9\n")
KERNEL(k_and_e64, "v_and_b32_e64 v20, v4, v9\n", "v_and_b32_e64 v21, v4, v9\n", "v_and_b32_e64 v22, v4, v9\n", "v_and_b32_e64 v23, v4, v9\n", "v_and_b32_e64 v24, v4, v9\n", "v_and_b32_e64 v25, v4, v9\n", "v_and_b32_e64 v26, v4, v9\n", "v_and_b32_e64 v27, v4, v9\n")
KERNEL(k_lshl_add_u64, "v_lshl_add_u64 v[20:21], v[4:5], 0, v[8:9]
", "v_lshl_add_u64 v[22:23], v[4:5], 0, v[8:9]
", "v_lshl_add_u64 v[24:25], v[4:5], 0, v[8:9]
", "v_lshl_add_u64 v[26:27], v[4:5], 0, v[8:9]
", "v_lshl_add_u64 v[20:21], v[6:7], 0, v[10:11]
", "v_lshl_add_u64 v[22:23], v[6:7], 0, v[10:11]
", "v_lshl_add_u64 v[24:25], v[6:7], 0, v[10:11]
", "v_lshl_add_u64 v[26:27], v[6:7], 0, v[10:11]
")

template <typename K>
static double run(K k, uint32_t* out, int blocks, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters); hipDeviceSynchronize();
    double best = 1e30;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(a); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate / 1e6; const int cus = p.multiProcessorCount;
    uint32_t* out; hipMalloc(&out, (size_t)cus * 64 * 256 * 4);
    const int iters = 60000;
    for (int wpc : {16, 32}) {
        const int blocks = cus * (wpc / 4);
        printf("-- %d waves/CU (%d per SIMD); cycles per wave-instruction per SIMD at the nominal %.2f GHz\n", wpc, wpc / 4, ghz);
#define SHOW(K) { double ms = run(K, out, blocks, iters); printf("%-12s %.3f ms -> %.2f cycles/instr\n", #K, ms, ms * 1e-3 * ghz * 1e9 / ((double)iters * 64 * (wpc / 4.0))); }
        SHOW(k_and_diff) SHOW(k_b3_same) SHOW(k_b3_diff) SHOW(k_b2c_diff) SHOW(k_and_same) SHOW(k_and_e64) SHOW(k_xor) SHOW(k_and_or) SHOW(k_or3) SHOW(k_bfi) SHOW(k_xad) SHOW(k_add3) SHOW(k_lshl_add_u64) SHOW(k_addc_same) SHOW(k_addc_diff) SHOW(k_and_diff)
    }
    return 0;
}
