// latency_probe.hip -- cycle costs of the building blocks of the front elimination on one wave of gfx950:
// hipcc --offload-arch=gfx950 -O3 -o latency_probe latency_probe.hip && ./latency_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double rcp2(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}
__device__ __forceinline__ double rdl(double x, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane), hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}
#define N 256
__global__ void k_probe(double *out, long long *cyc, int nw)
{
    const int lane = threadIdx.x & 63, l15 = lane & 15, l4 = lane >> 4;
    __shared__ double sh[4096];
    double x = 1.0 + 1e-3 * lane + out[0];
    long long t0, t1;
    int k = 0;
    // 0: dependent rcp chain
    t0 = clock64();
    for (int i = 0; i < N; ++i) x = rcp2(x) + 0.5;
    t1 = clock64(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
    // 1: dependent fma chain
    t0 = clock64();
    for (int i = 0; i < N; ++i) x = fma(x, 0.999, 0.001);
    t1 = clock64(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
    // 2: one dependent double shuffle (bpermute)
    t0 = clock64();
    for (int i = 0; i < N; ++i) x = __shfl(x, l15 + 16 * ((l4 + 1) & 3)) + 1e-9;
    t1 = clock64(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
    // 3: three independent shuffles + combine
    t0 = clock64();
    for (int i = 0; i < N; ++i) { const double a = __shfl(x, l15), b = __shfl(x, l15 + 16), c = __shfl(x, l15 + 32); x = fma(a, 0.3, fma(b, 0.3, fma(c, 0.3, 1e-9))); }
    t1 = clock64(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
    // 4: ten readlanes of a double feeding a dependent op
    t0 = clock64();
    for (int i = 0; i < N; ++i) {
        double s = 0;
#pragma unroll
        for (int j = 0; j < 10; ++j) s += rdl(x, 3 * j);
        x = s * 0.1 + 1e-9;
    }
    t1 = clock64(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
    // 5: dependent mfma chain
    d4 acc = {x, x, x, x};
    t0 = clock64();
    for (int i = 0; i < N; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, acc[0], acc, 0, 0, 0);
    t1 = clock64(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
    x = acc[0] + acc[3];
    // 6: LDS write + barrier + read (nw waves)
    t0 = clock64();
    for (int i = 0; i < N; ++i) { sh[threadIdx.x] = x; __syncthreads(); x = sh[(threadIdx.x + 17) % blockDim.x] + 1e-9; __syncthreads(); }
    t1 = clock64(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
    // 7: barrier alone
    t0 = clock64();
    for (int i = 0; i < N; ++i) __syncthreads();
    t1 = clock64(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
    // 8: global load dependent chain (pointer chase through L2)
    {
        long long *p = cyc + 64;
        long long idx = 0;
        t0 = clock64();
        for (int i = 0; i < N; ++i) idx = p[idx];
        t1 = clock64(); if (threadIdx.x == 0) cyc[k] = t1 - t0 + (idx & 0); ++k;
    }
    // 9: wall clock (100 MHz) vs shader clock over a fixed spin
    {
        const long long w0 = wall_clock64(); t0 = clock64();
        for (int i = 0; i < 4 * N; ++i) x = fma(x, 0.999, 0.001);
        t1 = clock64(); const long long w1 = wall_clock64();
        if (threadIdx.x == 0) { cyc[k] = t1 - t0; cyc[k + 1] = w1 - w0; } k += 2;
    }
    // 11: a scalar branch ladder (12 predicated bodies, wave-uniform conditions)
    {
        int sel = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        t0 = clock64();
        for (int i = 0; i < N; ++i) {
#pragma unroll
            for (int q = 0; q < 12; ++q) if (((sel + q + i) & 7) == 3) x = fma(x, 0.999, 0.001);
        }
        t1 = clock64(); if (threadIdx.x == 0) cyc[k] = t1 - t0; ++k;
    }
    out[threadIdx.x] = x;
}
int main()
{
    double *out; long long *cyc;
    hipMalloc(&out, 8 * 1024); hipMalloc(&cyc, 8 * 2048);
    hipMemset(out, 0, 8 * 1024);
    long long chase[1024];
    for (int i = 0; i < 1024; ++i) chase[i] = (i * 37 + 11) % 1024;
    const char *names[] = {"rcp2 + add (dependent)", "fma f64 (dependent)", "1 dependent shuffle (double)", "3 independent shuffles + 3 fma",
                           "10 readlane(double) + 10 adds", "mfma f64 16x16x4 (dependent)", "lds write + barrier + read + barrier", "barrier",
                           "global pointer chase (L2 hit)", "shader clocks of 1024 fma", "wall clocks (100 MHz) of the same", "12-way scalar branch ladder"};
    for (int nw : {1, 4, 8}) {
        hipMemset(cyc, 0, 8 * 2048);
        hipMemcpy(cyc + 64, chase, sizeof(chase), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(64 * nw), 0, 0, out, cyc, nw);
        hipDeviceSynchronize();
        long long h[16];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        printf("waves %d\n", nw);
        for (int k = 0; k < 12; ++k) printf("  %-40s %8.1f cycles per iteration\n", names[k], (double)h[k] / (k == 9 || k == 10 ? 1.0 : N));
        printf("  shader clock ~ %.0f MHz\n", 100.0 * h[9] / (double)h[10]);
    }
    return 0;
}
