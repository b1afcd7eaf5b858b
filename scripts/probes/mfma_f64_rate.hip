// Standalone probe: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 as a function of the number of
// independent accumulator chains per wave and of waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ void k(double *out, int iters, long long *clk)
{
    d4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1e-3;
    long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    double s = 0;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int CH>
void run(int wavesPerSimd, double *out, long long *clk)
{
    const int threads = 256 * wavesPerSimd > 1024 ? 1024 : 256 * wavesPerSimd;   // 4 SIMDs per CU
    const int blocksPerCU = (256 * wavesPerSimd) / threads;
    const int blocks = 256 * blocksPerCU;
    const int iters = 200000 / CH;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<CH><<<blocks, threads>>>(out, 1000, clk);
    hipDeviceSynchronize();
    float best = 1e30f;
    long long h[2];
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<CH><<<blocks, threads>>>(out, iters, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
    const double nmfma_per_simd = (double)iters * CH * wavesPerSimd;
    const double flops = (double)blocks * (threads / 64) * iters * CH * 2048.0;
    printf("chains %d waves/SIMD %d: %.2f ms  %.1f TFLOP/s  ns/MFMA/SIMD %.2f  memtime-ticks %lld realtime-ticks %lld\n", CH,
           wavesPerSimd, best, flops / best / 1e9, best * 1e6 / nmfma_per_simd, h[0], h[1]);
}

int main()
{
    double *out; long long *clk;
    hipMalloc(&out, sizeof(double) * 1024 * 2048);
    hipMalloc(&clk, 16);
    for (int w = 1; w <= 4; w *= 2) {
        run<1>(w, out, clk);
        run<2>(w, out, clk);
        run<4>(w, out, clk);
        run<8>(w, out, clk);
    }
    return 0;
}
