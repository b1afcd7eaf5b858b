// ldlt.hip -- batched dense LDL^T (no pivoting) and triangular solves for gfx950 (MI355X).
//
// This is the arithmetic the reference leaves to Ipopt's linear solver (MUMPS / MA57,
// /root/reference/examples/acopf/opf.jl:59-64): the factorisation of the interior-point KKT matrix
// K = [W J'; J -D] of each QP sub-problem (/root/reference/src/algorithms/subproblem_JuMP.jl:178).
//
// Layout: each instance owns a column-major Npad x Npad buffer (ld = Npad, Npad = 64*T); only the
// lower triangle is meaningful, rows/cols >= N are identity padding so every kernel works on whole
// 64x64 tiles.  Right-looking blocked algorithm, one panel of 64 columns per step k:
//   k_diag_factor : tile (k,k) -> L_kk, 1/D_k            one wave per instance (VALU, registers)
//   k_panel_trsm  : tiles (i,k), i>k -> L_ik, W_ik=L_ik D  one wave per tile  (VALU, registers)
//   k_trailing    : tiles (i,j), i>=j>k: A_ij -= W_ik L_jk'  v_mfma_f64_16x16x4_f64, LDS-staged
// The trailing update carries ~(1 - 3/(2T)) of the N^3/3 flops and is the kernel the roofline in
// bench.py is quoted on.
#include "sqphip_internal.hpp"

namespace sqphip {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------
// tile (k,k): unblocked LDL^T, thread r owns row r in registers, column broadcast through LDS
__global__ __launch_bounds__(64) void k_diag_factor(double *__restrict__ K, long strideK, int ld,
                                                   double *__restrict__ dinv, int Npad, int k,
                                                   const int *__restrict__ phase, int want)
{
    const int inst = blockIdx.x;
    if (phase && phase[inst] != want) return;
    const int r = threadIdx.x;
    double *A = K + (long)inst * strideK + (long)(k * 64) * ld + k * 64;   // tile origin
    __shared__ double colbuf[64];
    double a[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) a[c] = A[(long)c * ld + r];
    double my_dinv = 0.0;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        const double dj = __shfl(a[j], j);
        const double dji = 1.0 / dj;
        if (r == j) my_dinv = dji;
        const double l = a[j] * dji;          // L_rj for r > j
        __syncthreads();
        colbuf[r] = a[j];                     // d_j * L_rj  (row r of column j before scaling)
        __syncthreads();
        if (r > j) {
            a[j] = l;
#pragma unroll
            for (int c = j + 1; c < 64; ++c) a[c] -= l * colbuf[c];
        }
    }
#pragma unroll
    for (int c = 0; c < 64; ++c)
        if (c < r) A[(long)c * ld + r] = a[c];
    dinv[(long)inst * Npad + k * 64 + r] = my_dinv;
}

// ---------------------------------------------------------------------------------------------
// tiles (i,k), i>k: X L_kk' = A_ik by forward substitution along the row; W = X, L = X D^-1
__global__ __launch_bounds__(64) void k_panel_trsm(double *__restrict__ K, long strideK, int ld,
                                                  const double *__restrict__ dinv,
                                                  double *__restrict__ Wbuf, int Npad, int k,
                                                  const int *__restrict__ phase, int want)
{
    const int inst = blockIdx.y;
    if (phase && phase[inst] != want) return;
    const int i = k + 1 + blockIdx.x;
    const int r = threadIdx.x;
    double *Kb = K + (long)inst * strideK;
    const double *Lkk = Kb + (long)(k * 64) * ld + k * 64;
    double *A = Kb + (long)(k * 64) * ld + i * 64;
    __shared__ double Ls[64 * 64];   // Ls[c*64 + j] = L_kk[j][c]  (column c of the tile)
    __shared__ double dis[64];
#pragma unroll 8
    for (int c = 0; c < 64; ++c) Ls[c * 64 + r] = Lkk[(long)c * ld + r];
    dis[r] = dinv[(long)inst * Npad + k * 64 + r];
    double x[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) x[c] = A[(long)c * ld + r];
    __syncthreads();
#pragma unroll
    for (int j = 1; j < 64; ++j) {
        double s = x[j];
#pragma unroll
        for (int c = 0; c < j; ++c) s -= x[c] * Ls[c * 64 + j];
        x[j] = s;
    }
    double *W = Wbuf + (long)inst * Npad * 64;
#pragma unroll
    for (int c = 0; c < 64; ++c) {
        W[(long)c * Npad + i * 64 + r] = x[c];
        A[(long)c * ld + r] = x[c] * dis[c];
    }
}

// ---------------------------------------------------------------------------------------------
// trailing update on fp64 MFMA.  One workgroup (4 waves) per 64x64 tile (ti,tj), ti>=tj>k:
//   A[ti][tj] -= W[ti] * L[tj]'   with a 64-deep inner dimension.
// MFMA operands are arranged so that the accumulator's lane index runs along i (the contiguous
// direction of the column-major tile): T[jj][ii] = sum_k L[j][k] W[i][k], A-operand = L, B-operand = W.
// LDS images are k-major (Ls[k][j], Ws[k][i]) exactly as the columns lie in HBM; an XOR of 16 on
// the in-row index for odd k puts the two k-rows read by one 32-lane group on disjoint bank halves.
__device__ __forceinline__ int swz(int k, int i) { return k * 64 + (i ^ ((k & 1) << 4)); }

__global__ __launch_bounds__(256, 2) void k_trailing(double *__restrict__ K, long strideK, int ld,
                                                    const double *__restrict__ Wbuf, int Npad,
                                                    int T, int k, int B,
                                                    const int *__restrict__ phase, int want)
{
    const int rem = T - k - 1;
    const int ntl = rem * (rem + 1) / 2;
    int inst, t;
    const int bid = blockIdx.x;
    if ((B & 7) == 0) {             // keep one instance's tiles on one XCD (its panels stay in that L2)
        const int xcd = bid & 7, q = bid >> 3;
        inst = (q / ntl) * 8 + xcd;
        t = q % ntl;
    } else {
        inst = bid / ntl;
        t = bid % ntl;
    }
    if (phase && phase[inst] != want) return;
    // column-major enumeration of the lower triangle of the rem x rem tile grid
    int c = 0, cnt = rem;
    while (t >= cnt) { t -= cnt; ++c; --cnt; }
    const int tj = k + 1 + c, ti = tj + t;

    __shared__ double Ls[64 * 64];
    __shared__ double Ws[64 * 64];
    double *Kb = K + (long)inst * strideK;
    const double *Lg = Kb + (long)(k * 64) * ld + tj * 64;                 // L[tj] tile, ld
    const double *Wg = Wbuf + (long)inst * Npad * 64 + ti * 64;            // W[ti] tile, Npad
    const int tid = threadIdx.x;
    // stage both operand tiles: thread moves 2 doubles per (column) step; 32 threads cover a column
    {
        const int ii = (tid & 31) * 2, kk0 = tid >> 5;     // 8 columns per pass
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int kk = pass * 8 + kk0;
            const d2 lv = *reinterpret_cast<const d2 *>(Lg + (long)kk * ld + ii);
            const d2 wv = *reinterpret_cast<const d2 *>(Wg + (long)kk * Npad + ii);
            *reinterpret_cast<d2 *>(&Ls[swz(kk, ii)]) = lv;
            *reinterpret_cast<d2 *>(&Ws[swz(kk, ii)]) = wv;
        }
    }
    const int wave = tid >> 6, lane = tid & 63;
    const int jb = (wave >> 1) * 32, ib = (wave & 1) * 32;
    const int l15 = lane & 15, l4 = lane >> 4;
    double *Cg = Kb + (long)(tj * 64 + jb) * ld + ti * 64 + ib;
    // issue the C loads early: acc = C, then acc -= products (negated A operand)
    d4 acc[2][2];
#pragma unroll
    for (int bj = 0; bj < 2; ++bj)
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                acc[bj][bi][rr] = Cg[(long)(bj * 16 + l4 + 4 * rr) * ld + bi * 16 + l15];
    __syncthreads();
#pragma unroll 4
    for (int ks = 0; ks < 16; ++ks) {
        const int kk = ks * 4 + l4;
        const double a0 = -Ls[swz(kk, jb + l15)];
        const double a1 = -Ls[swz(kk, jb + 16 + l15)];
        const double b0 = Ws[swz(kk, ib + l15)];
        const double b1 = Ws[swz(kk, ib + 16 + l15)];
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
#pragma unroll
    for (int bj = 0; bj < 2; ++bj)
#pragma unroll
        for (int bi = 0; bi < 2; ++bi)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                Cg[(long)(bj * 16 + l4 + 4 * rr) * ld + bi * 16 + l15] = acc[bj][bi][rr];
}

// ---------------------------------------------------------------------------------------------
// forward substitution, step k.  Every workgroup of tile row i>=k re-solves the 64x64 unit-lower
// system of tile (k,k) (cheap), then either publishes v_k = D^-1 y_k (i==k) or updates x_i.
__global__ __launch_bounds__(64) void k_fwd_step(const double *__restrict__ K, long strideK, int ld,
                                                const double *__restrict__ dinv,
                                                double *__restrict__ x, double *__restrict__ v,
                                                int Npad, int k, const int *__restrict__ phase, int want)
{
    const int inst = blockIdx.y;
    if (phase && phase[inst] != want) return;
    const int i = k + blockIdx.x;
    const int r = threadIdx.x;
    const double *Kb = K + (long)inst * strideK;
    const double *Lkk = Kb + (long)(k * 64) * ld + k * 64;
    double *xb = x + (long)inst * Npad;
    __shared__ double ys[64];
    double lrow[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) lrow[c] = Lkk[(long)c * ld + r];
    double xr = xb[k * 64 + r];
#pragma unroll
    for (int j = 0; j < 63; ++j) {
        const double yj = __shfl(xr, j);
        if (r > j) xr -= lrow[j] * yj;
    }
    if (i == k) {
        v[(long)inst * Npad + k * 64 + r] = xr * dinv[(long)inst * Npad + k * 64 + r];
        return;
    }
    ys[r] = xr;
    __syncthreads();
    const double *Lik = Kb + (long)(k * 64) * ld + i * 64;
    double acc = 0.0;
#pragma unroll 16
    for (int c = 0; c < 64; ++c) acc += Lik[(long)c * ld + r] * ys[c];
    xb[i * 64 + r] -= acc;
}

// backward substitution L' x = v, step k (descending).  Workgroup j<=k re-solves the 64x64
// unit-upper system of tile (k,k)'; j==k publishes x_k, j<k updates v_j -= L_kj' x_k.
__global__ __launch_bounds__(64) void k_bwd_step(const double *__restrict__ K, long strideK, int ld,
                                                double *__restrict__ x, double *__restrict__ v,
                                                int Npad, int k, const int *__restrict__ phase, int want)
{
    const int inst = blockIdx.y;
    if (phase && phase[inst] != want) return;
    const int j = blockIdx.x;
    const int c = threadIdx.x;
    const double *Kb = K + (long)inst * strideK;
    const double *Lkk = Kb + (long)(k * 64) * ld + k * 64;
    double *vb = v + (long)inst * Npad;
    __shared__ double tile[64 * 65];
    __shared__ double xs[64];
    // stage tile (k,k) transposed-readable: tile[cc*65 + rr] = L[rr][cc]
#pragma unroll 8
    for (int cc = 0; cc < 64; ++cc) tile[cc * 65 + c] = Lkk[(long)cc * ld + c];
    __syncthreads();
    double xc = vb[k * 64 + c];
    for (int rr = 63; rr > 0; --rr) {
        const double xr = __shfl(xc, rr);
        if (c < rr) xc -= tile[c * 65 + rr] * xr;
    }
    if (j == k) {
        x[(long)inst * Npad + k * 64 + c] = xc;
        return;
    }
    xs[c] = xc;
    __syncthreads();
    const double *Lkj = Kb + (long)(j * 64) * ld + k * 64;   // tile (k,j): rows k-block, cols j-block
    __syncthreads();
#pragma unroll 8
    for (int cc = 0; cc < 64; ++cc) tile[cc * 65 + c] = Lkj[(long)cc * ld + c];
    __syncthreads();
    double acc = 0.0;
#pragma unroll 16
    for (int rr = 0; rr < 64; ++rr) acc += tile[c * 65 + rr] * xs[rr];
    vb[j * 64 + c] -= acc;
}

// ---------------------------------------------------------------------------------------------
void ldlt_factor(const LdltPlan &P, double *K, double *dinv, const int *phase, int want, Timers *tm)
{
    const long strideK = (long)P.ld * P.Npad;
    hipStream_t s = P.stream;
    for (int k = 0; k < P.T; ++k) {
        hipLaunchKernelGGL(k_diag_factor, dim3(P.B), dim3(64), 0, s, K, strideK, P.ld, dinv, P.Npad, k,
                           phase, want);
        const int rem = P.T - k - 1;
        if (rem <= 0) break;
        hipLaunchKernelGGL(k_panel_trsm, dim3(rem, P.B), dim3(64), 0, s, K, strideK, P.ld, dinv, P.Wbuf,
                           P.Npad, k, phase, want);
        const int ntl = rem * (rem + 1) / 2;
        std::pair<hipEvent_t, hipEvent_t> ev;
        const bool timed = tm && tm->enabled;
        if (timed) { ev = tm->get(); hipEventRecord(ev.first, s); }
        hipLaunchKernelGGL(k_trailing, dim3(ntl * P.B), dim3(256), 0, s, K, strideK, P.ld, P.Wbuf, P.Npad,
                           P.T, k, P.B, phase, want);
        if (timed) { hipEventRecord(ev.second, s); tm->pending_trailing.push_back(ev); }
        if (tm) tm->trailing_launches++;
    }
    if (tm) tm->n_factor++;
}

void ldlt_solve(const LdltPlan &P, const double *K, const double *dinv, double *x, double *v,
                const int *phase, int want)
{
    const long strideK = (long)P.ld * P.Npad;
    hipStream_t s = P.stream;
    for (int k = 0; k < P.T; ++k)
        hipLaunchKernelGGL(k_fwd_step, dim3(P.T - k, P.B), dim3(64), 0, s, K, strideK, P.ld, dinv, x, v,
                           P.Npad, k, phase, want);
    for (int k = P.T - 1; k >= 0; --k)
        hipLaunchKernelGGL(k_bwd_step, dim3(k + 1, P.B), dim3(64), 0, s, K, strideK, P.ld, x, v, P.Npad, k,
                           phase, want);
}

}  // namespace sqphip
