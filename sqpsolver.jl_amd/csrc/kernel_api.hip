// kernel_api.hip -- kernel-level C entry points (parity tests and micro-benchmarks of the LDL^T).
#include "../../include/sqphip.h"
#include "../../include/sqphip_test_hooks.h"
#include "sqphip_internal.hpp"
#include <cmath>
#include <cstring>
#include <random>

using namespace sqphip;

namespace {

__global__ void k_pad_identity(double *K, long strideK, int ld, int N, int Npad)
{
    const int inst = blockIdx.x;
    for (int i = N + threadIdx.x; i < Npad; i += blockDim.x) K[(long)inst * strideK + (long)i * ld + i] = 1.0;
}

__global__ void k_count_pos(const double *dinv, int Npad, int N, int *npos)
{
    const int inst = blockIdx.x;
    int c = 0;
    for (int i = threadIdx.x; i < N; i += blockDim.x) {
        const double d = dinv[(long)inst * Npad + i];
        if (d > 0.0 && isfinite(d)) ++c;
    }
    __shared__ int sh[256];
    sh[threadIdx.x] = c;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) npos[inst] = sh[0];
}

struct Scratch {
    LdltPlan P;
    double *K = nullptr, *dinv = nullptr, *x = nullptr, *v = nullptr;
    int *npos = nullptr;
    void alloc(int B, long N)
    {
        P.N = (int)N; P.Npad = (int)((N + 63) / 64 * 64); P.T = P.Npad / 64; P.ld = P.Npad; P.B = B;
        SQPHIP_HIP_OK(hipStreamCreate(&P.stream));
        SQPHIP_HIP_OK(hipMalloc(&K, sizeof(double) * (size_t)B * P.ld * P.Npad));
        SQPHIP_HIP_OK(hipMalloc(&P.Wbuf, sizeof(double) * (size_t)2 * LdltPlan::MAX_R * B * P.Npad * 64));
        P.init_lookahead();
        SQPHIP_HIP_OK(hipMalloc(&dinv, sizeof(double) * (size_t)B * P.Npad));
        SQPHIP_HIP_OK(hipMalloc(&x, sizeof(double) * (size_t)B * P.Npad));
        SQPHIP_HIP_OK(hipMalloc(&v, sizeof(double) * (size_t)B * P.Npad));
        SQPHIP_HIP_OK(hipMalloc(&npos, sizeof(int) * (size_t)B));
    }
    void upload(const double *A)
    {
        const long strideK = (long)P.ld * P.Npad;
        SQPHIP_HIP_OK(hipMemsetAsync(K, 0, sizeof(double) * (size_t)P.B * strideK, P.stream));
        for (int b = 0; b < P.B; ++b)
            SQPHIP_HIP_OK(hipMemcpy2DAsync(K + b * strideK, sizeof(double) * P.ld, A + (long)b * P.N * P.N,
                                           sizeof(double) * P.N, sizeof(double) * P.N, P.N,
                                           hipMemcpyHostToDevice, P.stream));
        hipLaunchKernelGGL(k_pad_identity, dim3(P.B), dim3(64), 0, P.stream, K, strideK, P.ld, P.N, P.Npad);
    }
    ~Scratch()
    {
        hipFree(K); hipFree(P.Wbuf); hipFree(dinv); hipFree(x); hipFree(v); hipFree(npos);
        P.destroy_lookahead();
        if (P.stream) hipStreamDestroy(P.stream);
    }
};

}  // namespace

extern "C" int sqphip_ldlt_factor_host(int32_t device, int32_t batch, int64_t N, double *A, double *dinv,
                                       int32_t *npos)
{
    try {
        SQPHIP_HIP_OK(hipSetDevice(device));
        Scratch S;
        S.alloc(batch, N);
        S.upload(A);
        ldlt_factor(S.P, S.K, S.dinv, nullptr, 0, nullptr);
        hipLaunchKernelGGL(k_count_pos, dim3(batch), dim3(256), 0, S.P.stream, S.dinv, S.P.Npad, S.P.N, S.npos);
        const long strideK = (long)S.P.ld * S.P.Npad;
        for (int b = 0; b < batch; ++b) {
            SQPHIP_HIP_OK(hipMemcpy2DAsync(A + (long)b * N * N, sizeof(double) * N, S.K + b * strideK,
                                           sizeof(double) * S.P.ld, sizeof(double) * N, N,
                                           hipMemcpyDeviceToHost, S.P.stream));
            SQPHIP_HIP_OK(hipMemcpyAsync(dinv + (long)b * N, S.dinv + (long)b * S.P.Npad, sizeof(double) * N,
                                         hipMemcpyDeviceToHost, S.P.stream));
        }
        SQPHIP_HIP_OK(hipMemcpyAsync(npos, S.npos, sizeof(int) * batch, hipMemcpyDeviceToHost, S.P.stream));
        SQPHIP_HIP_OK(hipStreamSynchronize(S.P.stream));
        return SQPHIP_OK;
    } catch (const std::string &e) {
        fprintf(stderr, "sqphip: %s\n", e.c_str());
        return SQPHIP_EHIP;
    }
}

extern "C" int sqphip_ldlt_solve_host(int32_t device, int32_t batch, int64_t N, const double *A, double *rhs)
{
    try {
        SQPHIP_HIP_OK(hipSetDevice(device));
        Scratch S;
        S.alloc(batch, N);
        S.upload(A);
        SQPHIP_HIP_OK(hipMemsetAsync(S.x, 0, sizeof(double) * (size_t)batch * S.P.Npad, S.P.stream));
        for (int b = 0; b < batch; ++b)
            SQPHIP_HIP_OK(hipMemcpyAsync(S.x + (long)b * S.P.Npad, rhs + (long)b * N, sizeof(double) * N,
                                         hipMemcpyHostToDevice, S.P.stream));
        // default: forward elimination fused into the factorisation (the product path); SQPHIP_FUSED_FWD=0
        // exercises the stand-alone forward steps (the refinement path)
        const char *ff = getenv("SQPHIP_FUSED_FWD");
        if (ff && ff[0] == '0') {
            ldlt_factor(S.P, S.K, S.dinv, nullptr, 0, nullptr);
            ldlt_solve(S.P, S.K, S.dinv, S.x, S.v, nullptr, 0);
        } else {
            ldlt_factor(S.P, S.K, S.dinv, nullptr, 0, nullptr, S.x, S.v);
            ldlt_solve(S.P, S.K, S.dinv, S.x, S.v, nullptr, 0, true);
        }
        for (int b = 0; b < batch; ++b)
            SQPHIP_HIP_OK(hipMemcpyAsync(rhs + (long)b * N, S.x + (long)b * S.P.Npad, sizeof(double) * N,
                                         hipMemcpyDeviceToHost, S.P.stream));
        SQPHIP_HIP_OK(hipStreamSynchronize(S.P.stream));
        return SQPHIP_OK;
    } catch (const std::string &e) {
        fprintf(stderr, "sqphip: %s\n", e.c_str());
        return SQPHIP_EHIP;
    }
}

namespace {
// diagonally dominant quasi-definite fill: K = [W J'; J -D], n1 = N*2/5 like the ACOPF shape
__global__ void k_fill_qd(double *K, long strideK, int ld, int N, int Npad, unsigned seed)
{
    const int inst = blockIdx.y;
    const int j = blockIdx.x;
    if (j >= Npad) return;
    double *col = K + (long)inst * strideK + (long)j * ld;
    const int n1 = N * 2 / 5;
    for (int i = j + threadIdx.x; i < Npad; i += blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761u) ^ (unsigned)(j * 40503u) ^ (seed + inst * 7919u);
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        double u = (double)(h & 0xFFFFFF) / 16777216.0 - 0.5;
        double val;
        if (i >= N || j >= N) val = (i == j) ? 1.0 : 0.0;
        else if (i == j) val = (j < n1 ? 1.0 : -1.0) * (0.05 * N + 1.0 + u);
        else val = 0.1 * u;
        col[i] = val;
    }
}
}  // namespace

extern "C" int sqphip_ldlt_bench(int32_t device, int32_t batch, int64_t N, int32_t reps,
                                 double *sec_per_factor, double *sec_trailing, int64_t *trailing_launches)
{
    try {
        SQPHIP_HIP_OK(hipSetDevice(device));
        Scratch S;
        S.alloc(batch, N);
        const long strideK = (long)S.P.ld * S.P.Npad;
        Timers tm;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        double total = 0.0;
        for (int rep = -1; rep < reps; ++rep) {
            hipLaunchKernelGGL(k_fill_qd, dim3(S.P.Npad, batch), dim3(128), 0, S.P.stream, S.K, strideK, S.P.ld,
                               S.P.N, S.P.Npad, 1234u + rep);
            tm.enabled = rep >= 0;
            hipEventRecord(e0, S.P.stream);
            ldlt_factor(S.P, S.K, S.dinv, nullptr, 0, &tm);
            hipEventRecord(e1, S.P.stream);
            SQPHIP_HIP_OK(hipEventSynchronize(e1));
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep >= 0) total += 1e-3 * ms; else tm.trailing_launches = 0;
        }
        tm.flush();
        *sec_per_factor = total / reps;
        *sec_trailing = tm.trailing_seconds / reps;
        *trailing_launches = tm.trailing_launches / reps;
        hipEventDestroy(e0); hipEventDestroy(e1);
        return SQPHIP_OK;
    } catch (const std::string &e) {
        fprintf(stderr, "sqphip: %s\n", e.c_str());
        return SQPHIP_EHIP;
    }
}

// ---------------------------------------------------------------------------------------------
// Stress test of the look-ahead schedule: the same random quasi-definite batch is factorised with the two-stream
// schedule and with everything on one stream (phase mask: a random subset of the instances is active); the
// factors must agree bit for bit.  Returns the number of repetitions with a mismatch.
namespace {
// counts entries that differ; ld > 0: the buffers are [.][ld x ld] column-major matrices and only the lower
// triangle (row >= column) is compared -- the strict upper triangle is never initialised nor read
__global__ void k_maxdiff(const double *a, const double *b, size_t n, int ld, unsigned long long *out)
{
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    unsigned long long bad = 0;
    for (; i < n; i += stride) {
        if (ld > 0) {
            const size_t e = i % ((size_t)ld * ld);
            if (e % ld < e / ld) continue;
        }
        const double x = a[i], y = b[i];
        if (!(x == y) && !(x != x && y != y)) ++bad;
    }
    if (bad) atomicAdd(out, bad);
}
}  // namespace

extern "C" int sqphip_ldlt_stress(int32_t device, int32_t batch, int64_t N, int32_t reps, int32_t *mismatches)
{
    try {
        SQPHIP_HIP_OK(hipSetDevice(device));
        Scratch S;
        S.alloc(batch, N);
        const long strideK = (long)S.P.ld * S.P.Npad;
        const size_t nk = (size_t)batch * strideK;
        double *K2, *dinv2; int *phase; unsigned long long *cnt;
        SQPHIP_HIP_OK(hipMalloc(&K2, sizeof(double) * nk));
        SQPHIP_HIP_OK(hipMalloc(&dinv2, sizeof(double) * (size_t)batch * S.P.Npad));
        SQPHIP_HIP_OK(hipMalloc(&phase, sizeof(int) * batch));
        SQPHIP_HIP_OK(hipMalloc(&cnt, sizeof(unsigned long long)));
        std::vector<int> hp(batch);
        int bad_reps = 0;
        hipStream_t aux = S.P.aux;
        S.P.lookahead_min = 0;         // this hook compares the two schedules whatever the matrix size
        for (int rep = 0; rep < reps; ++rep) {
            unsigned r = 12345u + 7919u * rep;
            for (int b = 0; b < batch; ++b) { r = r * 1664525u + 1013904223u; hp[b] = ((r >> 16) % 4) ? 1 : 0; }
            SQPHIP_HIP_OK(hipMemcpyAsync(phase, hp.data(), sizeof(int) * batch, hipMemcpyHostToDevice, S.P.stream));
            hipLaunchKernelGGL(k_fill_qd, dim3(S.P.Npad, batch), dim3(128), 0, S.P.stream, S.K, strideK, S.P.ld, S.P.N, S.P.Npad, 99u + rep);
            hipLaunchKernelGGL(k_fill_qd, dim3(S.P.Npad, batch), dim3(128), 0, S.P.stream, K2, strideK, S.P.ld, S.P.N, S.P.Npad, 99u + rep);
            // pivots of the instances the mask leaves out are never written: give both copies the same content
            SQPHIP_HIP_OK(hipMemsetAsync(S.dinv, 0, sizeof(double) * (size_t)batch * S.P.Npad, S.P.stream));
            SQPHIP_HIP_OK(hipMemsetAsync(dinv2, 0, sizeof(double) * (size_t)batch * S.P.Npad, S.P.stream));
            S.P.aux = aux;
            ldlt_factor(S.P, S.K, S.dinv, phase, 1, nullptr);
            S.P.aux = nullptr;
            ldlt_factor(S.P, K2, dinv2, phase, 1, nullptr);
            S.P.aux = aux;
            SQPHIP_HIP_OK(hipMemsetAsync(cnt, 0, sizeof(unsigned long long), S.P.stream));
            hipLaunchKernelGGL(k_maxdiff, dim3(2048), dim3(256), 0, S.P.stream, S.K, K2, nk, S.P.ld, cnt);
            hipLaunchKernelGGL(k_maxdiff, dim3(64), dim3(256), 0, S.P.stream, S.dinv, dinv2, (size_t)batch * S.P.Npad, 0, cnt);
            unsigned long long h = 0;
            SQPHIP_HIP_OK(hipMemcpyAsync(&h, cnt, sizeof h, hipMemcpyDeviceToHost, S.P.stream));
            SQPHIP_HIP_OK(hipStreamSynchronize(S.P.stream));
            if (h) { ++bad_reps; fprintf(stderr, "sqphip_ldlt_stress: rep %d: %llu entries differ\n", rep, h); }
        }
        *mismatches = bad_reps;
        hipFree(K2); hipFree(dinv2); hipFree(phase); hipFree(cnt);
        return SQPHIP_OK;
    } catch (const std::string &e) {
        fprintf(stderr, "sqphip: %s\n", e.c_str());
        return SQPHIP_EHIP;
    }
}

// ---------------------------------------------------------------------------------------------
// fp64 MFMA issue-rate probe: every wave runs `iters` x 4 independent v_mfma_f64_16x16x4_f64 from
// registers (no memory traffic).  Gives the on-box ceiling the LDL^T roofline is quoted against.
namespace {
typedef double d4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void k_mfma_f64_probe(double *out, int iters)
{
    d4v acc[4];
    const double a = 1e-3 * threadIdx.x, b = 1e-3;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = d4v{0.0, 0.0, 0.0, 0.0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
}  // namespace

extern "C" int sqphip_mfma_f64_peak(int32_t device, double *tflops)
{
    try {
        SQPHIP_HIP_OK(hipSetDevice(device));
        const int blocks = 256, threads = 1024, iters = 80000;   // 4 waves per SIMD, 4 chains per wave
        double *out;
        SQPHIP_HIP_OK(hipMalloc(&out, sizeof(double) * blocks * threads));
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_mfma_f64_probe, dim3(blocks), dim3(threads), 0, 0, out, 100);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_mfma_f64_probe, dim3(blocks), dim3(threads), 0, 0, out, iters);
        hipEventRecord(e1, 0);
        SQPHIP_HIP_OK(hipEventSynchronize(e1));
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = (double)blocks * (threads / 64) * (double)iters * 4.0 * 2.0 * 16 * 16 * 4;
        *tflops = flops / (1e-3 * ms) / 1e12;
        hipEventDestroy(e0); hipEventDestroy(e1);
        hipFree(out);
        return SQPHIP_OK;
    } catch (const std::string &e) {
        fprintf(stderr, "sqphip: %s\n", e.c_str());
        return SQPHIP_EHIP;
    }
}
