// Probe: cycles per step of the dependent chains the streamed top-of-tree solve is made of (one wave, gfx950).
//   a: y -= l * readlane(y, k)          (v_readlane_b32 x 2 -> v_fma_f64, each step depends on the previous one)
//   b: y -= l * c                        (v_fma_f64 chain alone)
//   c: eight independent readlanes, then eight fmas (the blocked form: pivots known before the block starts)
//   d: y -= l * ds_bpermute(y, k)        (__shfl)
// build: hipcc --offload-arch=gfx950 -O3 -o readlane_chain readlane_chain.hip ; run: ./readlane_chain
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ double rl(double x, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane), hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}
__global__ void k(double *out, long long *cyc, int n)
{
    const int lane = threadIdx.x;
    double y = 1.0 + 1e-3 * lane, l = 1e-6 * (lane + 1);
    long long t0 = clock64();
    for (int k = 0; k < n; ++k) y -= l * rl(y, k & 63);
    long long t1 = clock64();
    double z = y;
    for (int k = 0; k < n; ++k) z -= l * 0.999;
    long long t2 = clock64();
    double w = z;
    for (int k = 0; k < n; k += 8) {
        double p[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) p[q] = rl(w, (k + q) & 63);
#pragma unroll
        for (int q = 0; q < 8; ++q) w -= l * p[q];
    }
    long long t3 = clock64();
    double v = w;
    for (int k = 0; k < n; ++k) v -= l * __shfl(v, k & 63);
    long long t4 = clock64();
    // two registers per lane, as the kernel has them
    double a = v, b = v + 1.0;
    for (int k = 0; k < n; ++k) { const double yk = rl(a, k & 63); a -= l * yk; b -= l * yk; }
    long long t5 = clock64();
    out[lane] = a + b;
    if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; cyc[4] = t5 - t4; }
}
int main()
{
    double *o; long long *c, h[5];
    hipMalloc(&o, 64 * 8); hipMalloc(&c, 5 * 8);
    const int n = 4096;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, c, n);
        hipMemcpy(h, c, sizeof h, hipMemcpyDeviceToHost);
    }
    printf("cycles per step: readlane+fma %.1f | fma only %.1f | 8 readlanes then 8 fmas %.1f | shfl+fma %.1f | readlane + 2 fma %.1f\n",
           (double)h[0] / n, (double)h[1] / n, (double)h[2] / n, (double)h[3] / n, (double)h[4] / n);
    return 0;
}
