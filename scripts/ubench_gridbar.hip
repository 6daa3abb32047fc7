// ubench_gridbar.hip -- what does a device-wide barrier between the workgroups of ONE resident kernel cost on MI355X?
// (the alternative to one launch per dependent step: UPGMA / NJ merges, 11 us per launch).  Every spin is bounded.
// hipcc --offload-arch=gfx950 -O3 scripts/ubench_gridbar.hip -o /tmp/gb && /tmp/gb
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

// flat counter barrier: arrive (agent-scope release), spin on the counter (bounded), acquire
__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();
        atomicAdd(counter, 1u);
        int spins = 0;
        while (__atomic_load_n(counter, __ATOMIC_RELAXED) < target) {
            if (++spins > 2000000) { ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
    }
    __syncthreads();
    return ok;
}

// two-level: groups of 16 workgroups arrive at their own counter, the last arriver of a group at the top counter
__device__ __forceinline__ bool grid_barrier2(unsigned* group_counters, unsigned* top, unsigned it, unsigned n_groups, unsigned group_size_of_mine)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __threadfence();
        const unsigned g = blockIdx.x / 16;
        const unsigned prev = atomicAdd(&group_counters[g * 32], 1u); // 128-byte stride between the counters
        if (prev + 1 == (it + 1) * group_size_of_mine) atomicAdd(top, 1u);
        int spins = 0;
        while (__atomic_load_n(top, __ATOMIC_RELAXED) < (it + 1) * n_groups) {
            if (++spins > 2000000) { ok = false; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(256) void k_flat(unsigned* counter, float* data, int iters, int* err)
{
    float acc = 0;
    for (int it = 0; it < iters; ++it) {
        data[blockIdx.x * 256 + threadIdx.x] = acc + it;                  // something to publish
        if (!grid_barrier(counter, (unsigned)(it + 1) * gridDim.x)) { *err = 1; return; }
        acc += data[((blockIdx.x + 1) % gridDim.x) * 256 + threadIdx.x]; // read a neighbour's value of this step
    }
    data[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_two(unsigned* groups, unsigned* top, float* data, int iters, int* err)
{
    const unsigned n_groups = (gridDim.x + 15) / 16;
    const unsigned g = blockIdx.x / 16;
    const unsigned mine = g + 1 < n_groups ? 16u : gridDim.x - g * 16;
    float acc = 0;
    for (int it = 0; it < iters; ++it) {
        data[blockIdx.x * 256 + threadIdx.x] = acc + it;
        if (!grid_barrier2(groups, top, (unsigned)it, n_groups, mine)) { *err = 1; return; }
        acc += data[((blockIdx.x + 1) % gridDim.x) * 256 + threadIdx.x];
    }
    data[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ void k_empty(float* data) { data[threadIdx.x] += 1.f; }

int main()
{
    unsigned* ctr; float* data; int* err;
    hipMalloc(&ctr, 1 << 16); hipMalloc(&data, 2048 * 256 * 4); hipMalloc(&err, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 20000;
    for (int wgs : {64, 391, 1024}) {
        for (int variant = 0; variant < 2; ++variant) {
            hipMemset(ctr, 0, 1 << 16); hipMemset(data, 0, 2048 * 256 * 4); hipMemset(err, 0, 4);
            hipEventRecord(a);
            if (variant == 0) hipLaunchKernelGGL(k_flat, dim3(wgs), dim3(256), 0, 0, ctr, data, iters, err);
            else hipLaunchKernelGGL(k_two, dim3(wgs), dim3(256), 0, 0, ctr + 1024, ctr, data, iters, err);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            int e; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
            float v; hipMemcpy(&v, data, 4, hipMemcpyDeviceToHost);
            // acc after iters steps = sum of the neighbour's (acc + it): checks that the values crossed the barrier
            printf("%4d workgroups, %s barrier: %.2f us per step%s (data[0] = %g)\n", wgs, variant ? "two-level" : "flat", ms * 1e3 / iters,
                   e ? "  ** a spin ran out **" : "", v);
        }
    }
    hipEventRecord(a);
    for (int i = 0; i < 20000; ++i) hipLaunchKernelGGL(k_empty, dim3(391), dim3(256), 0, 0, data);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("for comparison, 20000 dependent launches of a 391-workgroup kernel: %.2f us per launch\n", ms * 1e3 / 20000);
    return 0;
}
