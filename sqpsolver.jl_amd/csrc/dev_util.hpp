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
// Occupancy target of the per-sweep vector kernels (second __launch_bounds__ argument, waves per SIMD): 4 = one workgroup
// of 1024 threads per CU with up to 128 registers per lane.  8 (two workgroups per CU, 64 registers) was measured in
// round 3: the fused stages then spill 160 - 400 bytes per lane and 512 x IEEE-118 loses 8 % (5407 -> 4970 QP/s).
#ifndef SQPHIP_VEC_WAVES_PER_EU
#define SQPHIP_VEC_WAVES_PER_EU 4
#endif

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

// several reductions behind ONE pair of barriers (a block_reduce costs two workgroup barriers, ~2 300 cycles each way
// with sixteen waves behind global stores: six in a row were 14 000 cycles of the prepare stage).  Same operations on
// the same operands in the same order as the single reductions: the results are bit-identical.
template <class O0, class O1, class O2, class O3, class O4, class O5>
static __device__ __forceinline__ void block_reduce6(double &v0, double &v1, double &v2, double &v3, double &v4, double &v5)
{
    __shared__ double sh6[6][TPB / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        v0 = O0::f(v0, __shfl_xor(v0, o)); v1 = O1::f(v1, __shfl_xor(v1, o)); v2 = O2::f(v2, __shfl_xor(v2, o));
        v3 = O3::f(v3, __shfl_xor(v3, o)); v4 = O4::f(v4, __shfl_xor(v4, o)); v5 = O5::f(v5, __shfl_xor(v5, o));
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        sh6[0][w] = v0; sh6[1][w] = v1; sh6[2][w] = v2; sh6[3][w] = v3; sh6[4][w] = v4; sh6[5][w] = v5;
    }
    __syncthreads();
    v0 = sh6[0][0]; v1 = sh6[1][0]; v2 = sh6[2][0]; v3 = sh6[3][0]; v4 = sh6[4][0]; v5 = sh6[5][0];
#pragma unroll
    for (int w = 1; w < TPB / 64; ++w) {
        v0 = O0::f(v0, sh6[0][w]); v1 = O1::f(v1, sh6[1][w]); v2 = O2::f(v2, sh6[2][w]);
        v3 = O3::f(v3, sh6[3][w]); v4 = O4::f(v4, sh6[4][w]); v5 = O5::f(v5, sh6[5][w]);
    }
}
template <class O0, class O1, class O2, class O3, class O4, class O5, class O6>
static __device__ __forceinline__ void block_reduce7(double &v0, double &v1, double &v2, double &v3, double &v4, double &v5, double &v6)
{
    __shared__ double sh7[7][TPB / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        v0 = O0::f(v0, __shfl_xor(v0, o)); v1 = O1::f(v1, __shfl_xor(v1, o)); v2 = O2::f(v2, __shfl_xor(v2, o));
        v3 = O3::f(v3, __shfl_xor(v3, o)); v4 = O4::f(v4, __shfl_xor(v4, o)); v5 = O5::f(v5, __shfl_xor(v5, o));
        v6 = O6::f(v6, __shfl_xor(v6, o));
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        const int w = threadIdx.x >> 6;
        sh7[0][w] = v0; sh7[1][w] = v1; sh7[2][w] = v2; sh7[3][w] = v3; sh7[4][w] = v4; sh7[5][w] = v5; sh7[6][w] = v6;
    }
    __syncthreads();
    v0 = sh7[0][0]; v1 = sh7[1][0]; v2 = sh7[2][0]; v3 = sh7[3][0]; v4 = sh7[4][0]; v5 = sh7[5][0]; v6 = sh7[6][0];
#pragma unroll
    for (int w = 1; w < TPB / 64; ++w) {
        v0 = O0::f(v0, sh7[0][w]); v1 = O1::f(v1, sh7[1][w]); v2 = O2::f(v2, sh7[2][w]);
        v3 = O3::f(v3, sh7[3][w]); v4 = O4::f(v4, sh7[4][w]); v5 = O5::f(v5, sh7[5][w]); v6 = O6::f(v6, sh7[6][w]);
    }
}
template <class O0, class O1>
static __device__ __forceinline__ void block_reduce2(double &v0, double &v1)
{
    __shared__ double sh2[2][TPB / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { v0 = O0::f(v0, __shfl_xor(v0, o)); v1 = O1::f(v1, __shfl_xor(v1, o)); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sh2[0][threadIdx.x >> 6] = v0; sh2[1][threadIdx.x >> 6] = v1; }
    __syncthreads();
    v0 = sh2[0][0]; v1 = sh2[1][0];
#pragma unroll
    for (int w = 1; w < TPB / 64; ++w) { v0 = O0::f(v0, sh2[0][w]); v1 = O1::f(v1, sh2[1][w]); }
}

static __device__ __forceinline__ bool fin(double v) { return isfinite(v); }

// Inertia correction: the shift tried after `dw` has given the wrong inertia (Ipopt's schedule; dw_last = last shift
// that worked in this solve).  k_inertia walks this sequence; the sparse path factorises TWO consecutive members per
// sweep whenever the first is not a sure thing (mf_speculates): a failed first shift then costs no extra sweep.
static __device__ __forceinline__ double next_shift(double dw, double dw_last)
{
    if (dw == 0.0) return dw_last == 0.0 ? 1e-4 : fmax(1e-20, dw_last / 3.0);
    return dw * (dw_last == 0.0 ? 100.0 : 8.0);
}
// a second candidate is worth its flops when the first shift is a shrink attempt or a retry (it fails about one time
// in three); the plain delta_w = 0 of a solve that never needed a correction is not speculated on
static __device__ __forceinline__ bool mf_speculates(const DV &d, const IpmState &st)
{
    return d.sparse && d.mf.fronts1 != nullptr && d.spec_mode != 0 && (st.fac_attempt > 0 || (d.spec_mode == 1 && st.dw > 0.0));
}

}  // namespace sqphip
