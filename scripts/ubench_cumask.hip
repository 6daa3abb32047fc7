// ubench_cumask.hip -- which CUs a CU-masked stream's workgroups land on (round 6: the CU-partition experiment of
// DESIGN 4.7).  Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/ubench_cumask scripts/ubench_cumask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>

__global__ void where_kernel(unsigned* out, int spin)
{
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));
        const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); // HW_ID whole
        out[blockIdx.x] = (xcc << 24) | (hw & 0xffffff);
    }
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
}

static void run(const char* name, hipStream_t st, int wgs)
{
    unsigned* d;
    hipMalloc(&d, wgs * 4);
    hipLaunchKernelGGL(where_kernel, dim3(wgs), dim3(256), 0, st, d, 20000); // 0.2 ms each: all resident at once
    std::vector<unsigned> h(wgs);
    hipMemcpyAsync(h.data(), d, wgs * 4, hipMemcpyDeviceToHost, st);
    hipStreamSynchronize(st);
    std::map<unsigned, std::set<unsigned>> per_xcc;
    for (unsigned v : h) {
        const unsigned xcc = v >> 24, cu = (v >> 8) & 0xf, sh = (v >> 12) & 1, se = (v >> 13) & 7;
        per_xcc[xcc].insert((se << 5) | (sh << 4) | cu);
    }
    printf("%s: %d workgroups\n", name, wgs);
    size_t total = 0;
    for (auto& kv : per_xcc) {
        printf("  xcc %u: %zu distinct (se,sh,cu):", kv.first, kv.second.size());
        for (unsigned c : kv.second) printf(" %u.%u.%u", c >> 5, (c >> 4) & 1, c & 15);
        printf("\n");
        total += kv.second.size();
    }
    printf("  = %zu CUs\n", total);
    hipFree(d);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s: %d CUs\n", p.gcnArchName, cus);
    hipStream_t plain;
    hipStreamCreate(&plain);
    run("unmasked", plain, 2048);
    auto masked = [&](const char* name, int from, int to, int step, int keep) {
        std::vector<uint32_t> m((cus + 31) / 32, 0u);
        for (int i = from; i < to; ++i)
            if (step == 0 || (i % step) < keep) m[i / 32] |= 1u << (i % 32);
        hipStream_t st;
        if (hipExtStreamCreateWithCUMask(&st, (uint32_t)m.size(), m.data()) != hipSuccess) {
            printf("%s: hipExtStreamCreateWithCUMask failed\n", name);
            return;
        }
        run(name, st, 2048);
        hipStreamDestroy(st);
    };
    masked("bits [0, 64)", 0, 64, 0, 0);
    masked("bits [64, 256)", 64, cus, 0, 0);
    masked("bits i % 8 < 2", 0, cus, 8, 2);
    masked("bits [0, 32)", 0, 32, 0, 0);
    return 0;
}
