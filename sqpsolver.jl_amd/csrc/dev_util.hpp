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
    return d.sparse && d.mf.fronts1 != nullptr && (st.fac_attempt > 0 || (d.spec_mode == 1 && st.dw > 0.0));
}

}  // namespace sqphip
