// ubench_xcd.hip -- a chain of dependent steps kept on ONE XCD of the MI355X.
//
// A device-wide barrier between steps costs 4.4 / 20 / 45 us with 64 / 391 / 1024 workgroups (ubench_gridbar.hip): what costs
// is the agent-scope release / acquire -- with ordinary (coarse-grained) device memory the eight XCDs' L2 caches are only
// made coherent by writing back and invalidating.  Workgroups that all run on the SAME XCD share one L2: a barrier
// between them is an atomic at that L2, stores become visible to the others once they have left the CU (the vector L1
// is write-through; s_waitcnt vmcnt(0)), and readers only have to bypass their own L1 (sc1 loads) -- no fence wider
// than the workgroup is needed.  This measures that barrier and CHECKS the visibility claim: every step, every
// workgroup publishes a value and a 64-float row, and every workgroup reads all of them back.
//
// Workgroups are picked by where they really run: each reads the hardware register XCC_ID; those on XCD 0 take a ticket,
// the first P of them take part, everything else exits at once.
//
// hipcc --offload-arch=gfx950 -O3 scripts/ubench_xcd.hip -o /tmp/ux && /tmp/ux
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ unsigned xcc_id()
{
    // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20), offset 0, size 32: simm16 = (size-1) << 11 | offset << 6 | id
    return (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xFu;
}

__device__ __forceinline__ unsigned ld_u32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_f32(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// all P participants arrive; no fence beyond the workgroup's own stores having left the CU
__device__ __forceinline__ bool xcd_barrier(unsigned* counter, unsigned target)
{
    __builtin_amdgcn_s_waitcnt(0); // vmcnt(0) expcnt(0) lgkmcnt(0): this lane's stores are acknowledged by the L2
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (ld_u32(counter) < target)
            if (++spins > 4000000) { ok = false; break; }
    }
    __syncthreads();
    return ok;
}

// WT (mode 3): the writers use agent-scope relaxed atomic STORES (sc1: written through the XCD's L2) -- with sc1 loads on the
// other side that should be coherent between XCDs without any fence, which plain stores are not
// mode 0: participants = the first P ticket holders on XCD `want_xcd`;  mode 1: participants = blocks 0..P-1 wherever they run
// FENCE: the readers use PLAIN loads and an agent-scope ACQUIRE fence after the barrier (invalidates the CU's L1) instead
// of sc1 loads -- what a kernel with many loads would rather do
template <bool FENCE, bool WT = false>
__global__ __launch_bounds__(256) void k_chain(unsigned* ctl, unsigned* slots, float* rows, int P, int iters, int mode, unsigned want_xcd,
                                               unsigned* stats)
{
    __shared__ unsigned s_rank;
    const unsigned x = xcc_id();
    if (threadIdx.x == 0) {
        unsigned r = ~0u;
        if (mode == 0) {
            if (x == want_xcd) r = atomicAdd(&ctl[0], 1u);
        } else
            r = blockIdx.x;
        if (r < (unsigned)P) atomicAdd(&stats[8 + x], 1u); // where the participants run
        s_rank = r;
    }
    __syncthreads();
    const unsigned rank = s_rank;
    if (rank >= (unsigned)P) return;
    unsigned errors = 0;
    for (int it = 0; it < iters; ++it) {
        if (WT) {
            if (threadIdx.x == 0) __hip_atomic_store(&slots[rank * 32], (unsigned)it + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (threadIdx.x < 64) __hip_atomic_store(&rows[rank * 64 + threadIdx.x], (float)(it + 1) + 0.5f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            if (threadIdx.x == 0) slots[rank * 32] = (unsigned)it + 1u;                 // 128-byte stride
            if (threadIdx.x < 64) rows[rank * 64 + threadIdx.x] = (float)(it + 1) + 0.5f; // a plain store
        }
        if (!xcd_barrier(&ctl[32], (unsigned)(it + 1) * 2u * P - P)) { atomicAdd(&stats[0], 1u); return; }
        if (FENCE) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            for (int w = threadIdx.x; w < P; w += 256)
                if (((volatile unsigned*)slots)[w * 32] != (unsigned)it + 1u) ++errors;
            for (int e = threadIdx.x; e < P * 64; e += 256)
                if (((volatile float*)rows)[e] != (float)(it + 1) + 0.5f) ++errors;
        } else {
            for (int w = threadIdx.x; w < P; w += 256)
                if (ld_u32(&slots[w * 32]) != (unsigned)it + 1u) ++errors;
            for (int e = threadIdx.x; e < P * 64; e += 256)
                if (ld_f32(&rows[e]) != (float)(it + 1) + 0.5f) ++errors;
        }
        // second barrier: nobody overwrites its slot for step it+1 while another workgroup still reads step it
        if (!xcd_barrier(&ctl[32], (unsigned)(it + 1) * 2u * P)) { atomicAdd(&stats[0], 1u); return; }
    }
    if (errors) atomicAdd(&stats[1], errors);
}

int main()
{
    unsigned *ctl, *slots, *stats;
    float* rows;
    hipMalloc(&ctl, 4096); hipMalloc(&slots, 1024 * 128); hipMalloc(&rows, 1024 * 64 * 4); hipMalloc(&stats, 256);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 20000;
    // mode 2: any XCD, but the shared words live in UNCACHED (fine-grained) device memory: coherent between the XCDs
    // without write-back / invalidate, at the price of every access going to memory
    unsigned *u_ctl, *u_slots;
    float* u_rows;
    hipExtMallocWithFlags((void**)&u_ctl, 4096, hipDeviceMallocUncached);
    hipExtMallocWithFlags((void**)&u_slots, 256 * 128, hipDeviceMallocUncached);
    hipExtMallocWithFlags((void**)&u_rows, 256 * 64 * 4, hipDeviceMallocUncached);
    for (int mode = 0; mode < 3; ++mode)
        for (int P : {8, 16, 32, 64, 128, 256}) {
            if (mode == 0 && P > 64) continue;
            if (mode == 2) {
                hipMemset(u_ctl, 0, 4096); hipMemset(u_slots, 0, 256 * 128); hipMemset(u_rows, 0, 256 * 64 * 4); hipMemset(stats, 0, 256);
                hipEventRecord(a);
                hipLaunchKernelGGL(k_chain<false>, dim3(P), dim3(256), 0, 0, u_ctl, u_slots, u_rows, P, iters, 1, 0u, stats);
                hipEventRecord(b);
                hipError_t e = hipEventSynchronize(b);
                float ms = 0; hipEventElapsedTime(&ms, a, b);
                unsigned h[64]; hipMemcpy(h, stats, 256, hipMemcpyDeviceToHost);
                printf("any XCD, UNCACHED mem  P=%3d: %.2f us per step, timeouts %u, stale reads %u  [%s]\n", P, 1e3 * ms / iters, h[0], h[1], hipGetErrorString(e));
                continue;
            }
            hipMemset(ctl, 0, 4096); hipMemset(slots, 0, 256 * 128); hipMemset(rows, 0, 256 * 64 * 4); hipMemset(stats, 0, 256);
            const int grid = mode == 0 ? 8 * (P + 16) : P; // enough workgroups for XCD 0 to receive P of them
            hipEventRecord(a);
            hipLaunchKernelGGL(k_chain<false>, dim3(grid), dim3(256), 0, 0, ctl, slots, rows, P, iters, mode, 0u, stats);
            hipEventRecord(b);
            hipError_t e = hipEventSynchronize(b);
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            unsigned h[64]; hipMemcpy(h, stats, 256, hipMemcpyDeviceToHost);
            printf("%s P=%2d: %.2f us per step (2 barriers + %d sc1 loads per lane), timeouts %u, stale reads %u, participants per XCD:",
                   mode == 0 ? "one XCD (by XCC_ID)  " : "any XCD (blocks 0..P)", P, 1e3 * ms / iters, (P + 255) / 256 + (P * 64 + 255) / 256, h[0], h[1]);
            for (int x = 0; x < 8; ++x) printf(" %u", h[8 + x]);
            printf("  [%s]\n", hipGetErrorString(e));
        }
    for (int P : {16, 32, 64}) { // one XCD, plain loads behind an acquire fence
        hipMemset(ctl, 0, 4096); hipMemset(slots, 0, 256 * 128); hipMemset(rows, 0, 256 * 64 * 4); hipMemset(stats, 0, 256);
        hipEventRecord(a);
        hipLaunchKernelGGL(k_chain<true>, dim3(8 * (P + 16)), dim3(256), 0, 0, ctl, slots, rows, P, iters, 0, 0u, stats);
        hipEventRecord(b);
        hipError_t e = hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        unsigned h[64]; hipMemcpy(h, stats, 256, hipMemcpyDeviceToHost);
        printf("one XCD, plain loads + acquire fence P=%2d: %.2f us per step, timeouts %u, stale reads %u  [%s]\n", P, 1e3 * ms / iters, h[0], h[1], hipGetErrorString(e));
    }
    for (int P : {8, 16, 32, 64, 128, 256, 512, 1024}) { // any XCD, write-through stores + sc1 loads, no fence
        hipMemset(ctl, 0, 4096); hipMemset(slots, 0, 1024 * 128); hipMemset(rows, 0, 1024 * 64 * 4); hipMemset(stats, 0, 256);
        hipEventRecord(a);
        hipLaunchKernelGGL((k_chain<false, true>), dim3(P), dim3(256), 0, 0, ctl, slots, rows, P, iters, 1, 0u, stats);
        hipEventRecord(b);
        hipError_t e = hipEventSynchronize(b);
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        unsigned h[64]; hipMemcpy(h, stats, 256, hipMemcpyDeviceToHost);
        printf("any XCD, sc1 STORES + sc1 loads, no fence P=%4d: %.2f us per step, timeouts %u, stale reads %u  [%s]\n", P, 1e3 * ms / iters, h[0], h[1], hipGetErrorString(e));
    }
    return 0;
}
