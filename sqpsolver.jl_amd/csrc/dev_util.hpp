// dev_util.hpp -- workgroup-level reductions shared by the vector kernels
#pragma once
#include <hip/hip_runtime.h>

namespace sqphip {

// threads of the per-instance vector kernels (interior-point vectors, merit reductions, ACOPF callbacks): these are
// latency-bound gather loops over O(n + m + nnz) items of ONE instance per workgroup, so the lever is waves in flight
// per instance.  Measured on MI355X, 512 x IEEE-118: 256 -> 1024 threads, see DESIGN.md section 5.
#ifndef SQPHIP_VEC_THREADS
#define SQPHIP_VEC_THREADS 1024
#endif
#define TPB SQPHIP_VEC_THREADS

struct OpSum { __device__ static double f(double a, double b) { return a + b; } };
struct OpMax { __device__ static double f(double a, double b) { return fmax(a, b); } };
struct OpMin { __device__ static double f(double a, double b) { return fmin(a, b); } };

// wave64 butterfly (every lane ends with the wave's value), then a 4-entry LDS exchange
template <class Op> static __device__ __forceinline__ double block_reduce(double v)
{
    __shared__ double sh[TPB / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = Op::f(v, __shfl_xor(v, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = sh[0];
#pragma unroll
    for (int w = 1; w < TPB / 64; ++w) r = Op::f(r, sh[w]);
    return r;
}

static __device__ __forceinline__ bool fin(double v) { return isfinite(v); }

}  // namespace sqphip
