// Probe: what a dependent global round trip costs a 1024-thread workgroup that owns one "instance" (a 1.2 MB region) out
// of hundreds, as the vector stages of the sweep do -- and whether touching the region first (one discarded load per
// 128-byte line, all in flight at once) turns the later dependent loads into cache hits.
//   hipcc --offload-arch=gfx950 -O3 -o roundtrip_probe roundtrip_probe.hip ; ./roundtrip_probe [instances] [region doubles]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ __launch_bounds__(1024) void k(const double *base, long region, int nseg, int touch, double *out, long long *cyc)
{
    const double *p = base + (long)blockIdx.x * region;
    const int tid = threadIdx.x;
    double acc = 0.0;
    long long t0 = clock64();
    if (touch) {
        // one load per 128-byte line of the part of the region the segments will read; results never used
        const long lines = (long)nseg * 2048 / 16;
        for (long l = tid; l < lines; l += 1024) {
            int tmp;
            asm volatile("global_load_dword %0, %1, off" : "=v"(tmp) : "v"(p + l * 16));
        }
    }
    long long t1 = clock64();
    // nseg dependent segments: each reads 2048 fresh doubles (two per thread), the address of the second depends on the first
    int idx = tid;
    for (int s = 0; s < nseg; ++s) {
        const double a = p[(long)s * 2048 + idx];
        const int j = (int)(a * 0.0) + (1023 - tid);            // data dependence, same index set
        const double b = p[(long)s * 2048 + 1024 + j];
        acc += a + b;
        __syncthreads();
    }
    long long t2 = clock64();
    out[(long)blockIdx.x * 1024 + tid] = acc;
    if (tid == 0) { cyc[2 * blockIdx.x] = t1 - t0; cyc[2 * blockIdx.x + 1] = t2 - t1; }
}
int main(int argc, char **argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 128;
    const long region = argc > 2 ? atol(argv[2]) : 150000;      // doubles per instance (1.2 MB)
    const int nseg = 32;
    double *buf, *out; long long *cyc;
    hipMalloc(&buf, sizeof(double) * region * B); hipMemset(buf, 0, sizeof(double) * region * B);
    hipMalloc(&out, sizeof(double) * 1024 * B); hipMalloc(&cyc, sizeof(long long) * 2 * B);
    // something that evicts the caches between runs: a 1 GB fill
    double *junk; hipMalloc(&junk, 1L << 30);
    std::vector<long long> h(2 * B);
    for (int touch = 0; touch < 2; ++touch)
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(junk, rep, 1L << 30);
            hipLaunchKernelGGL(k, dim3(B), dim3(1024), 0, 0, buf, region, nseg, touch, out, cyc);
            hipMemcpy(h.data(), cyc, sizeof(long long) * 2 * B, hipMemcpyDeviceToHost);
            double a = 0, b = 0;
            for (int i = 0; i < B; ++i) { a += h[2 * i]; b += h[2 * i + 1]; }
            printf("instances %d touch %d rep %d: touch phase %.0f cycles, %d dependent segments %.0f cycles = %.0f per segment (two dependent loads each)\n",
                   B, touch, rep, a / B, nseg, b / B, b / B / nseg);
        }
    return 0;
}
