// ubench_gridbar.hip -- what a meeting of all resident workgroups costs on an MI355X (round 6: neighbour joining as one
// resident launch).  Cooperative launch of G workgroups x 1024 threads, K meetings, four ways:
//   0  one counter: release fetch_add, acquire-load spin                (what cooperative groups do)
//   1  one counter: fences outside, relaxed spin
//   2  eight counters (blockIdx & 7) + one: the last of eight adds to the shared one
//   3  no read-modify-write at all: every workgroup stores its tagged slot, every workgroup polls all G slots
//   4  3 without the release / acquire fences (a meeting that orders control only)
// Build: hipcc --offload-arch=gfx950 -O2 -o famsa_amd/_build/ubench_gridbar scripts/ubench_gridbar.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define AGENT __HIP_MEMORY_SCOPE_AGENT

__global__ __launch_bounds__(1024) void bar_kernel(uint32_t* ctr, uint64_t* slots, int K, int variant, long long* cycles)
{
    const int G = gridDim.x, w = blockIdx.x, tid = threadIdx.x;
    const long long t0 = wall_clock64();
    for (int k = 1; k <= K; ++k) {
        __syncthreads();
        if (variant == 0) {
            if (tid == 0) {
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, AGENT);
                while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, AGENT) < (uint32_t)k * G) __builtin_amdgcn_s_sleep(1);
            }
        } else if (variant == 1) {
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, AGENT);
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, AGENT) < (uint32_t)k * G) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        } else if (variant == 2) {
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                uint32_t* sub = ctr + 16 * (1 + (w & 7));
                const int members = (G - (w & 7) + 7) / 8;
                const uint32_t old = __hip_atomic_fetch_add(sub, 1u, __ATOMIC_RELAXED, AGENT);
                if ((int)(old + 1) == k * members) __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, AGENT);
                const uint32_t groups = G < 8 ? G : 8;
                while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, AGENT) < (uint32_t)k * groups) __builtin_amdgcn_s_sleep(1);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        } else {
            uint64_t* buf = slots + (size_t)(k & 1) * G;
            if (tid == 0) {
                if (variant == 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_store(buf + w, ((uint64_t)k << 32) | (uint32_t)w, __ATOMIC_RELAXED, AGENT);
            }
            for (;;) {
                int ok = 1;
                for (int t = tid; t < G; t += 1024)
                    if ((__hip_atomic_load(buf + t, __ATOMIC_RELAXED, AGENT) >> 32) != (uint64_t)k) ok = 0;
                if (__syncthreads_and(ok)) break;
            }
            if (variant == 3 && tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        __syncthreads();
    }
    if (w == 0 && tid == 0) *cycles = wall_clock64() - t0;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs, wall clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, 100000);
    uint32_t* ctr;
    uint64_t* slots;
    long long* cyc;
    hipMalloc(&ctr, 4096);
    hipMalloc(&slots, 2 * 1024 * 8);
    hipMalloc(&cyc, 8);
    const int K = 2000;
    for (int variant = 0; variant < 5; ++variant)
        for (int G : {8, 32, 64, 128, 256}) {
            if (G > p.multiProcessorCount) continue;
            hipMemset(ctr, 0, 4096);
            hipMemset(slots, 0, 2 * 1024 * 8);
            int k = K, v = variant;
            void* args[] = {&ctr, &slots, &k, &v, &cyc};
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            hipError_t e = hipLaunchCooperativeKernel((const void*)bar_kernel, dim3(G), dim3(1024), args, 0, 0);
            hipEventRecord(e1, 0);
            if (e != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
                printf("variant %d G %d: %s\n", variant, G, hipGetErrorString(e));
                return 1;
            }
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            long long c = 0;
            hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("variant %d  G %3d  %7.2f us per meeting (wall clock: %7.2f us)\n", variant, G, ms * 1000.0 / K,
                   (double)c / 100.0 / K);
        }
    return 0;
}
