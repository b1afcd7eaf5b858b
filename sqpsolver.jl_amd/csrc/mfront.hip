// mfront.hip -- batched multifrontal LDL^T (no pivoting) of the sparse Newton matrix and its triangular solves,
// for gfx950 (MI355X).
//
// This is the arithmetic the reference leaves to Ipopt's sparse linear solver (MUMPS / MA57,
// /root/reference/examples/acopf/opf.jl:59-64) behind JuMP.optimize! (/root/reference/src/algorithms/
// subproblem_JuMP.jl:178): numeric factorisation of the interior-point Newton matrix of every QP sub-problem, once
// per interior-point iteration, and 2-4 solves with the factors.
//
// Plan (mfplan.hip / symbolic.hip, built once per sparsity structure): supernodes in postorder, one dense front per
// supernode in the instance's front arena, fronts grouped by level of the assembly tree.  Numeric phase, level by
// level from the leaves:
//   k_mf_factor   one workgroup per (front, instance): zero the front (in LDS for fronts up to 90 rows), gather the
//                 structural entries of the Newton matrix from the sub-problem data (item lists: every destination
//                 summed in a fixed order), extend-add the contribution blocks of the children (one child after the
//                 other), eliminate the supernode's columns (right-looking rank-1 updates, one barrier per column),
//                 write L and the contribution block back.  An extra last ROW of the front carries the right-hand
//                 side through the same elimination: the forward solve of the first right-hand side costs nothing.
//   k_mf_fwd      further right-hand sides: per front y = b_cols + sum of the children's updates, triangular solve
//                 with L11, update vector for the ancestors (stored where the contribution block's last row lives).
//   k_mf_bwd      levels in reverse: x_cols = L11^-T (D^-1 y - L21' x_rows), x_rows gathered from the finished ancestors.
// No atomics anywhere: every entry has one writer per launch and every sum a fixed order, so a batch gives the same
// bits on every run.  HBM traffic per factorisation and instance: the item data (12 B per structural entry),
// 8 nnz(L) written, the contribution blocks written once and read once.
#include "ctx.hpp"
#include "dev_util.hpp"
#include <cstdlib>

namespace sqphip {

#define MF_REG_P 1e-8      // = IPM_REG_P / IPM_REG_D of ipm.hip
#define MF_REG_D 1e-8

__device__ __forceinline__ double mf_item_value(const MfItem &it, const double *hv, const double *jv, const double *Dd,
                                                const double *sigp, const double *hd, const int *rt, double hsc,
                                                double dw)
{
    switch (it.type) {
    case MF_ITEM_H: return hsc * hv[it.a];
    case MF_ITEM_JKEPT: return rt[it.row] != ROW_FREE ? jv[it.a] : 0.0;
    case MF_ITEM_PAIR: return rt[it.row] != ROW_FREE ? jv[it.a] * jv[it.b] / (Dd[it.row] + MF_REG_D) : 0.0;
    case MF_ITEM_VDIAG: return hd[it.a] + sigp[it.a] + dw + MF_REG_P;
    default: return rt[it.row] != ROW_FREE ? -(Dd[it.row] + MF_REG_D) : -1.0;
    }
}

// Thread layout inside a front: RL row lanes x (NT / RL) column groups; RL = 16 / 32 / 64 by front height so that a
// tiny front does not idle three quarters of its lanes.
template <int NT, bool INPLACE>
__global__ __launch_bounds__(NT) void k_mf_factor(DV d, int sbegin, int want, int with_rhs)
{
    const int inst = blockIdx.y;
    if (d.phase[inst] != want) return;
    const MfDev &M = d.mf;
    const int s = M.sched[sbegin + blockIdx.x];
    const int nc = M.nc[s], nr = M.nr[s], fs = nc + nr, ld = fs + 1, f0 = M.first[s];
    double *G = M.fronts + (long)inst * M.stride + M.off[s];
    extern __shared__ double mf_lds[];
    double *F = INPLACE ? G : mf_lds;
    const int tid = threadIdx.x;
    const int rl_log = ld <= 16 ? 4 : (ld <= 32 ? 5 : 6);
    const int RL = 1 << rl_log, rlane = tid & (RL - 1), cg = tid >> rl_log, ncg = NT >> rl_log;
    // 1. zero
    for (int e = tid; e < ld * fs; e += NT) F[e] = 0.0;
    __syncthreads();
    // 2. structural entries of the Newton matrix that live in this front, right-hand side row
    {
        const IpmState &st = d.ist[inst];
        const double hsc = st.hsc, dw = st.dw;
        const double *hv = d.hv + (long)inst * d.nnzhc, *jv = d.jv + (long)inst * d.nnzjc;
        const double *Dd = d.Dd + (long)inst * d.m, *sigp = d.sigp + (long)inst * d.n, *hd = d.hd + (long)inst * d.n;
        const int *rt = d.rtype + (long)inst * d.m;
        for (int e = M.asm_ptr[s] + tid; e < M.asm_ptr[s + 1]; e += NT) {
            double a = 0.0;
            for (int k = M.item_ptr[e]; k < M.item_ptr[e + 1]; ++k) a += mf_item_value(M.items[k], hv, jv, Dd, sigp, hd, rt, hsc, dw);
            F[M.dest_loc[e]] = a;
        }
        if (with_rhs) {
            const double *b = d.xv + (long)inst * d.Fpad + f0;
            for (int j = tid; j < nc; j += NT) F[j * ld + fs] = b[j];
        }
    }
    __syncthreads();
    // 3. extend-add of the children's contribution blocks (rows cnc.. of columns cnc.. of the child's front, its
    //    right-hand-side row included), one child after the other: two children may hit the same entry
    for (int q = M.child_ptr[s]; q < M.child_ptr[s + 1]; ++q) {
        const int c = M.child[q];
        const int cnc = M.nc[c], cnr = M.nr[c], cld = cnc + cnr + 1;
        const double *Cg = M.fronts + (long)inst * M.stride + M.off[c] + (long)cnc * cld + cnc;
        const int *rel = M.rel + M.rowptr[c];
        for (int jj = cg; jj < cnr; jj += ncg) {
            const int gj = rel[jj] * ld;
            const double *Cc = Cg + (long)jj * cld;
            for (int ii = jj + rlane; ii <= cnr; ii += RL) F[gj + (ii < cnr ? rel[ii] : fs)] += Cc[ii];
        }
        __syncthreads();
    }
    // 4. eliminate the nc columns of the supernode: right-looking, one barrier per column.  Column k keeps the
    //    unscaled entries d_k L_ik (nobody writes it after step k - 1); L_jk = F_jk / d_k is formed on the fly.
    for (int k = 0; k < nc; ++k) {
        const double *Fk = F + k * ld;
        const double di = 1.0 / Fk[k];
        for (int j = k + 1 + cg; j < fs; j += ncg) {
            const double lj = Fk[j] * di;
            double *Fj = F + j * ld;
            for (int i = j + rlane; i <= fs; i += RL) Fj[i] -= Fk[i] * lj;
        }
        __syncthreads();
    }
    // 5. results: L (scaled) and 1 / D, z = D^-1 L^-1 b, contribution block with its right-hand-side row
    double *dinv = d.dinv + (long)inst * d.Fpad + f0, *vv = d.vv + (long)inst * d.Fpad + f0;
    for (int j = cg; j < fs; j += ncg) {
        const double *Fj = F + j * ld;
        double *Gj = G + (long)j * ld;
        if (j < nc) {
            const double di = 1.0 / Fj[j];
            for (int i = j + 1 + rlane; i < fs; i += RL) Gj[i] = Fj[i] * di;
            if (rlane == 0) { dinv[j] = di; if (with_rhs) vv[j] = Fj[fs] * di; }
        } else if (!INPLACE) {
            for (int i = j + rlane; i <= fs; i += RL) Gj[i] = Fj[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------

// forward substitution of one front by one wave: x (d.xv) holds the right-hand side in elimination order; on return
// d.vv holds D^-1 L^-1 b for the columns of the front and the update for the ancestors sits in the last row of the
// front's contribution block
__global__ __launch_bounds__(64) void k_mf_fwd(DV d, int sbegin, int want, int generic)
{
    const int inst = blockIdx.y;
    if (d.phase[inst] != want) return;
    const MfDev &M = d.mf;
    const int s = M.sched[sbegin + blockIdx.x];
    const int nc = M.nc[s], nr = M.nr[s], fs = nc + nr, ld = fs + 1, f0 = M.first[s];
    double *G = M.fronts + (long)inst * M.stride + M.off[s];
    const int lane = threadIdx.x;
    extern __shared__ double mf_lds[];
    double *y = mf_lds;                       // fs entries (the launch sizes it for its largest front)
    const double *b = d.xv + (long)inst * d.Fpad + f0;
    for (int i = lane; i < fs; i += 64) y[i] = i < nc ? b[i] : 0.0;
    __syncthreads();
    for (int q = M.child_ptr[s]; q < M.child_ptr[s + 1]; ++q) {
        const int c = M.child[q];
        const int cnc = M.nc[c], cnr = M.nr[c], cfs = cnc + cnr, cld = cfs + 1;
        const double *Cg = M.fronts + (long)inst * M.stride + M.off[c];
        const int *rel = M.rel + M.rowptr[c];
        for (int jj = lane; jj < cnr; jj += 64) y[rel[jj]] += Cg[(long)(cnc + jj) * cld + cfs];
        __syncthreads();
    }
    const double *dinv = d.dinv + (long)inst * d.Fpad + f0;
    double *vv = d.vv + (long)inst * d.Fpad + f0;
    if (nc <= 64 && !generic) {
        // lane i owns y_i of the triangular part; y_k travels by a wave shuffle (no barrier, loads pipeline freely)
        double yi = lane < nc ? y[lane] : 0.0;
#pragma unroll 4
        for (int k = 0; k < nc - 1; ++k) {
            const double l = (lane > k && lane < nc) ? G[(long)k * ld + lane] : 0.0;
            const double yk = __shfl(yi, k);
            yi -= l * yk;
        }
        if (lane < nc) { y[lane] = yi; vv[lane] = yi * dinv[lane]; }
        __syncthreads();
        for (int i = nc + lane; i < fs; i += 64) {
            double acc = y[i];
#pragma unroll 4
            for (int k = 0; k < nc; ++k) acc -= G[(long)k * ld + i] * y[k];
            G[(long)i * ld + fs] = acc;
        }
    } else {
        for (int k = 0; k < nc; ++k) {
            const double yk = y[k];
            for (int i = k + 1 + lane; i < fs; i += 64) y[i] -= G[(long)k * ld + i] * yk;
            __syncthreads();
        }
        for (int k = lane; k < nc; k += 64) vv[k] = y[k] * dinv[k];
        for (int i = nc + lane; i < fs; i += 64) G[(long)i * ld + fs] = y[i];
    }
}

// backward substitution of one front by one wave: x_cols = L11^-T (vv_cols - L21' x_rows) into d.xv
__global__ __launch_bounds__(64) void k_mf_bwd(DV d, int sbegin, int want, int generic)
{
    const int inst = blockIdx.y;
    if (d.phase[inst] != want) return;
    const MfDev &M = d.mf;
    const int s = M.sched[sbegin + blockIdx.x];
    const int nc = M.nc[s], nr = M.nr[s], fs = nc + nr, ld = fs + 1, f0 = M.first[s];
    const double *G = M.fronts + (long)inst * M.stride + M.off[s];
    const int lane = threadIdx.x;
    extern __shared__ double mf_lds[];
    double *x = mf_lds;
    double *xg = d.xv + (long)inst * d.Fpad;
    const double *vv = d.vv + (long)inst * d.Fpad + f0;
    const int *rows = M.rows + M.rowptr[s];
    for (int i = lane; i < fs; i += 64) x[i] = i < nc ? vv[i] : xg[rows[i - nc]];
    __syncthreads();
    if (nc <= 64 && !generic) {
        // lane k owns column k: its own dot product with x_rows, then the unit upper triangular solve by shuffles
        double t = lane < nc ? x[lane] : 0.0;
        if (lane < nc) {
            const double *Gk = G + (long)lane * ld;
#pragma unroll 4
            for (int i = nc; i < fs; ++i) t -= Gk[i] * x[i];
        }
#pragma unroll 4
        for (int i = nc - 1; i > 0; --i) {
            const double l = lane < i ? G[(long)lane * ld + i] : 0.0;
            const double xi = __shfl(t, i);
            t -= l * xi;
        }
        if (lane < nc) xg[f0 + lane] = t;
    } else {
        for (int k = nc - 1; k >= 0; --k) {
            double a = 0.0;
            for (int i = k + 1 + lane; i < fs; i += 64) a += G[(long)k * ld + i] * x[i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
            if (lane == 0) x[k] -= a;
            __syncthreads();
        }
        for (int k = lane; k < nc; k += 64) xg[f0 + k] = x[k];
    }
}

// ---------------------------------------------------------------------------------------------------------------
static int mf_generic_solves()
{
    static const int g = getenv("SQPHIP_MF_GENERIC") ? atoi(getenv("SQPHIP_MF_GENERIC")) : 0;
    return g;
}

void mf_factor(Ctx &C, int want, bool with_rhs)
{
    const DV &d = C.d;
    hipStream_t s = C.stream;
    for (const MfLaunch &L : C.mfp.fac) {
        const dim3 grid(L.count, d.B);
        if (L.threads == 64)
            hipLaunchKernelGGL((k_mf_factor<64, false>), grid, dim3(64), L.lds_bytes, s, d, L.begin, want, (int)with_rhs);
        else if (L.lds_bytes > 0)
            hipLaunchKernelGGL((k_mf_factor<256, false>), grid, dim3(256), L.lds_bytes, s, d, L.begin, want, (int)with_rhs);
        else
            hipLaunchKernelGGL((k_mf_factor<256, true>), grid, dim3(256), 0, s, d, L.begin, want, (int)with_rhs);
    }
    C.mf_factor_launches += (long)C.mfp.fac.size();
}

// x (d.xv) <- K^-1 x through the factors; skip_fwd: d.vv already holds D^-1 L^-1 b (fused into mf_factor)
void mf_solve(Ctx &C, int want, bool skip_fwd)
{
    const DV &d = C.d;
    hipStream_t s = C.stream;
    const int generic = mf_generic_solves();
    if (!skip_fwd)
        for (const MfLaunch &L : C.mfp.fwd)
            hipLaunchKernelGGL(k_mf_fwd, dim3(L.count, d.B), dim3(64), L.lds_bytes, s, d, L.begin, want, generic);
    for (const MfLaunch &L : C.mfp.bwd)
        hipLaunchKernelGGL(k_mf_bwd, dim3(L.count, d.B), dim3(64), L.lds_bytes, s, d, L.begin, want, generic);
}

}  // namespace sqphip
