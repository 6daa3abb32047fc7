// ubench_lds.hip -- what does it cost 8 waves of one workgroup to stream an LDS array of float4 (the staging area of
// clarans_eval_kernel) -- as ds_read_b128 (array of structures) against three ds_read_b32 (structure of arrays), with and
// without the ballot / compacted ds_write_b128 that follows in the kernel?  dev tool.
// hipcc --offload-arch=gfx950 -O3 scripts/ubench_lds.hip -o /tmp/ul && /tmp/ul
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, int klo, int khi)
{
    __shared__ float4 s_e[1024];
    __shared__ float s_x[1024], s_y[1024];
    __shared__ int s_z[1024];
    __shared__ float4 s_we[8][128 + 16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 1024; i += blockDim.x) {
        const int nn = (i * 7) % 100;
        s_e[i] = make_float4(1.0f + i, (i % 37) ? 0.0f : -1.0f, __int_as_float(nn), 0.0f);
        s_x[i] = 1.0f + i; s_y[i] = (i % 37) ? 0.0f : -1.0f; s_z[i] = nn;
    }
    __syncthreads();
    const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    float acc = 0.0f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (MODE == 6) { // the same work, software-pipelined: the next pass's entries and the next batch of the walk are
                     // requested before the current ones are consumed
        for (int it = 0; it < iters; ++it) {
            asm volatile("" ::: "memory");
            float4 e[2], en[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) e[u] = s_e[64 * u + lane];
            for (int s0 = 0; s0 < 1024; s0 += 128) {
                const int sn = s0 + 128 < 1024 ? s0 + 128 : s0;
#pragma unroll
                for (int u = 0; u < 2; ++u) en[u] = s_e[sn + 64 * u + lane]; // in flight during this pass
                int m = 0;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int nn = __float_as_int(e[u].z);
                    const bool mine = e[u].y < 0.0f || (nn >= klo && nn < khi);
                    const uint64_t mask = __ballot(mine);
                    if (mine) s_we[wave][m + __popcll(mask & lt_mask)] = e[u];
                    m += __popcll(mask);
                }
                const int mpad = (m + 7) & ~7;
                if (lane < mpad - m + 8) s_we[wave][m + lane] = make_float4(0.f, 0.f, __int_as_float(-1), 0.f); // + one batch to prefetch into
                __builtin_amdgcn_wave_barrier();
                float4 f[8], g[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) f[u] = s_we[wave][u];
                for (int i = 0; i < mpad; i += 8) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) g[u] = s_we[wave][i + 8 + u];
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc += lane == __float_as_int(f[u].z) ? f[u].x : f[u].y;
#pragma unroll
                    for (int u = 0; u < 8; ++u) f[u] = g[u];
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int u = 0; u < 2; ++u) e[u] = en[u];
            }
        }
    } else
    for (int it = 0; it < iters; ++it) {
        for (int s0 = 0; s0 < 1024; s0 += 128) {
            asm volatile("" ::: "memory"); // the array is read again in every pass
            float4 e[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = s0 + 64 * u + lane;
                if (MODE != 1 && MODE != 5) e[u] = s_e[t];
                else e[u] = make_float4(s_x[t], s_y[t], __int_as_float(s_z[t]), 0.0f);
            }
            if (MODE >= 2) { // + the compaction
                int m = 0;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int nn = __float_as_int(e[u].z);
                    const bool mine = e[u].y < 0.0f || (nn >= klo && nn < khi);
                    const uint64_t mask = __ballot(mine);
                    if (mine) s_we[wave][m + __popcll(mask & lt_mask)] = e[u];
                    m += __popcll(mask);
                }
                __builtin_amdgcn_wave_barrier();
                if (MODE >= 3) { // the kernel's walk: batches of 8; MODE 4 / 5: only the lanes that own a slot take part
                    const int mpad = (m + 7) & ~7;
                    if (lane < mpad - m) s_we[wave][m + lane] = make_float4(0.f, 0.f, __int_as_float(-1), 0.f);
                    __builtin_amdgcn_wave_barrier();
                    if (MODE == 3 || lane < 16)
                        for (int i = 0; i < mpad; i += 8) {
                            float4 f[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) f[u] = s_we[wave][i + u];
#pragma unroll
                            for (int u = 0; u < 8; ++u) acc += lane == __float_as_int(f[u].z) ? f[u].x : f[u].y;
                        }
                } else
                    acc += (float)m;
                __builtin_amdgcn_wave_barrier();
            } else {
                acc += e[0].x + e[1].y + e[0].z;
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + tid] = acc;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main()
{
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1024 * 512 * 4); hipMalloc(&cyc, 1024 * 8);
    const int iters = 200;
    for (int blocks : {1, 1024})
        for (int mode = 0; mode < 7; ++mode) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, out, cyc, iters, 0, 13);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, out, cyc, iters, 0, 13);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(512), 0, 0, out, cyc, iters, 0, 13);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(512), 0, 0, out, cyc, iters, 0, 13);
            if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(512), 0, 0, out, cyc, iters, 0, 13);
            if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(512), 0, 0, out, cyc, iters, 0, 13);
            if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(512), 0, 0, out, cyc, iters, 0, 13);
            hipDeviceSynchronize();
            unsigned long long h = 0; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            const char* names[] = {"float4 (b128) read only", "3 x b32 (SoA) read only", "float4 read + compaction", "float4 read + compaction + walk (8s)", "... the walk by 16 lanes only", "... and the reads as 3 x b32", "compaction + walk, software-pipelined"};
            printf("%4d workgroups, %-34s: %.0f clocks per 128-entry pass of one wave (8 waves share the array)\n", blocks, names[mode],
                   (double)h / iters / 8);
        }
    // fewer waves per workgroup, more slots per wave: the own entries are walked once whatever the split, the negative
    // ones and the padding once per wave, and a SIMD then runs one wave's compaction instead of two
    for (int waves : {8, 4, 2}) {
        hipLaunchKernelGGL(k<3>, dim3(1024), dim3(64 * waves), 0, 0, out, cyc, iters, 0, 100 / waves + 1);
        hipDeviceSynchronize();
        unsigned long long h = 0; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        printf("1024 workgroups of %d waves, each wave owning %d of 100 slots: %.0f clocks per 128-entry pass\n", waves, 100 / waves + 1,
               (double)h / iters / 8);
    }
    return 0;
}
