// ordered_sum_check.hip -- famsa_amd/csrc/ordered_sum.h against one float addition after the other on the host, bit for
// bit, over vectors made to hit every branch: ties, binade changes, negative / huge / non-finite / denormal addends.
// Built by __graft_entry__.build() into famsa_amd/_build/ordered_sum_check; run by tests/test_gpu_ordered_sum.py.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../famsa_amd/csrc/ordered_sum.h"

constexpr int MAX_N = 5000;

__global__ __launch_bounds__(64) void sum_kernel(const float* x, const int* off, const int* len, float* out, uint32_t* stats,
                                                 long long* ticks)
{
    __shared__ __attribute__((aligned(16))) float row[lcsgpu::ordered_sum_padded(MAX_N)]; // as the callers have it: in LDS
    const int n = len[blockIdx.x];
    for (int t = threadIdx.x; t < lcsgpu::ordered_sum_padded(n); t += 64) row[t] = x[off[blockIdx.x] + t];
    __syncthreads();
    uint32_t st[3] = {0, 0, 0};
    const long long t0 = wall_clock64();
    const float s = lcsgpu::wave_ordered_sum(row, n, st);
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[blockIdx.x] = s;
        for (int k = 0; k < 3; ++k) stats[3 * blockIdx.x + k] = st[k];
        ticks[blockIdx.x] = t1 - t0;
    }
}

__global__ __launch_bounds__(64) void seq_kernel(const float* x, const int* off, const int* len, float* out, long long* ticks)
{
    __shared__ __attribute__((aligned(16))) float row[lcsgpu::ordered_sum_padded(MAX_N)];
    const int n = len[blockIdx.x];
    for (int t = threadIdx.x; t < lcsgpu::ordered_sum_padded(n); t += 64) row[t] = x[off[blockIdx.x] + t];
    __syncthreads();
    const long long t0 = wall_clock64();
    float s = 0.0f;
    for (int i = 0; i < n; i += 8) { // (n padded with +0.0f)
        float g[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) g[q] = row[i + q];
#pragma unroll
        for (int q = 0; q < 8; ++q) s = __fadd_rn(s, g[q]);
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[blockIdx.x] = s;
        ticks[blockIdx.x] = t1 - t0;
    }
}

static float host_sum(const float* x, int n)
{
    volatile float s = 0.0f; // one rounding per addition, no reassociation
    for (int i = 0; i < n; ++i) s = s + x[i];
    return s;
}

int main(int argc, char** argv)
{
    std::mt19937 rng(20260601);
    std::vector<float> x;
    std::vector<int> off, len, kind;
    auto uni = [&](double a, double b) { return (float)std::uniform_real_distribution<double>(a, b)(rng); };
    const int n_cases = argc > 1 ? atoi(argv[1]) : 6000; // (a few hundred: one wave per CU, the time of a lone wave)
    for (int c = 0; c < n_cases; ++c) {
        const int k = c % 8;
        int n = (int)(rng() % MAX_N) + 1;
        if (c % 97 == 0) n = (int)(rng() % 70) + 1;
        off.push_back((int)x.size());
        len.push_back(n);
        kind.push_back(k);
        const size_t base = x.size();
        x.resize(base + lcsgpu::ordered_sum_padded(n), 0.0f);
        float* v = x.data() + base;
        for (int i = 0; i < n; ++i) {
            switch (k) {
            case 0: v[i] = uni(0.0, 1.0); break;                                     // plain
            case 1: v[i] = (float)(rng() % 65536) / 8192.0f; break;                  // quantised: ties every few additions
            case 2: v[i] = uni(-0.3, 1.0); break;                                    // some negative
            case 3: v[i] = uni(0.4, 3.5); break;                                     // like distances
            case 4: v[i] = (rng() % 7 == 0) ? 0.0f : (float)(rng() % 4096) / 1024.0f * (float)std::ldexp(1.0, (int)(rng() % 5) - 2); break;
            case 5: v[i] = std::ldexp(uni(0.5, 1.0), (int)(rng() % 40) - 20); break; // wide range of magnitudes
            case 6: v[i] = (rng() % 3 == 0) ? std::ldexp(1.0f, -140 + (int)(rng() % 20)) : uni(0.0, 1e-30); break; // denormals, tiny
            default: v[i] = uni(0.0, 2.0); break;
            }
        }
        if (k == 7 && n > 10) { // rare specials
            const int what = (int)(rng() % 5), at = (int)(rng() % n);
            if (what == 0) v[at] = INFINITY;
            else if (what == 1) v[at] = NAN;
            else if (what == 2) v[at] = 3.0e38f;
            else if (what == 3) v[at] = -0.0f;
            else v[at] = -1.0e6f;
        }
    }
    float *dx, *dout;
    int *doff, *dlen;
    uint32_t* dstats;
    long long* dticks;
    hipMalloc(&dstats, n_cases * 12);
    hipMalloc(&dticks, n_cases * 8);
    hipMalloc(&dx, x.size() * 4);
    hipMalloc(&dout, n_cases * 4);
    hipMalloc(&doff, n_cases * 4);
    hipMalloc(&dlen, n_cases * 4);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(doff, off.data(), n_cases * 4, hipMemcpyHostToDevice);
    hipMemcpy(dlen, len.data(), n_cases * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(sum_kernel, dim3(n_cases), dim3(64), 0, 0, dx, doff, dlen, dout, dstats, dticks);
    std::vector<float> out(n_cases);
    if (hipMemcpy(out.data(), dout, n_cases * 4, hipMemcpyDeviceToHost) != hipSuccess) {
        printf("FAIL: the kernel did not run\n");
        return 2;
    }
    std::vector<uint32_t> stats(3 * n_cases);
    std::vector<long long> ticks(n_cases);
    hipMemcpy(stats.data(), dstats, n_cases * 12, hipMemcpyDeviceToHost);
    hipMemcpy(ticks.data(), dticks, n_cases * 8, hipMemcpyDeviceToHost);
    for (int k = 0; k < 8; ++k) {
        unsigned long long a = 0, b = 0, q = 0, elems = 0;
        long long tk = 0;
        for (int c = k; c < n_cases; c += 8) { a += stats[3 * c]; b += stats[3 * c + 1]; q += stats[3 * c + 2]; elems += len[c]; tk += ticks[c]; }
        printf("kind %d: %llu blocks of 256 at once, %llu pieces of 64 at once, %llu pieces one by one; %.2f ns per addend\n", k, a, b, q,
               (double)tk * 10.0 / (double)elems);
    }
    {
        hipLaunchKernelGGL(seq_kernel, dim3(n_cases), dim3(64), 0, 0, dx, doff, dlen, dout, dticks);
        std::vector<long long> tk(n_cases);
        hipMemcpy(tk.data(), dticks, n_cases * 8, hipMemcpyDeviceToHost);
        long long all = 0, elems = 0;
        for (int c = 0; c < n_cases; ++c) { all += tk[c]; elems += len[c]; }
        printf("one addition after the other on the device: %.2f ns per addend\n", (double)all * 10.0 / (double)elems);
    }
    int bad = 0;
    for (int c = 0; c < n_cases; ++c) {
        const float h = host_sum(x.data() + off[c], len[c]);
        uint32_t a, b;
        memcpy(&a, &h, 4);
        memcpy(&b, &out[c], 4);
        if (a != b && !(std::isnan(h) && std::isnan(out[c]))) {
            if (bad < 10) printf("case %d kind %d n %d: host %.9g (%08x) device %.9g (%08x)\n", c, kind[c], len[c], h, a, out[c], b);
            ++bad;
        }
    }
    printf("%s: %d of %d sums differ\n", bad ? "FAIL" : "OK", bad, n_cases);
    return bad ? 1 : 0;
}
