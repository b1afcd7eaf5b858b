// mf_dev.hpp -- device helpers of the multifrontal path shared by mfront.hip (the kernels) and ipm.hip (the stage kernel
// that assembles the matrix values behind the Newton right-hand side).  Not part of the ABI.
#pragma once
#include "ctx.hpp"
#include "dev_util.hpp"

namespace sqphip {

// Two candidates per sweep: 0 = the factorisation with the shift st.dw, 1 = the one with the next shift of the schedule
// (dev_util.hpp next_shift), computed in the same launches for the instances mf_speculates() names; k_inertia picks
// (IpmState.sel) and the solves follow.  The factor kernels run over 2 B "instances": blockIdx.y >= B is candidate 1.
__device__ __forceinline__ double *mf_arena(const DV &d, int inst, int cand) { return (cand ? d.mf.fronts1 : d.mf.fronts) + (long)inst * d.mf.stride; }
__device__ __forceinline__ double *mf_vals(const DV &d, int inst, int cand) { return (cand ? d.mf.vals1 : d.mf.vals) + (long)inst * d.mf.nnzK; }
__device__ __forceinline__ double *mf_dinv(const DV &d, int inst, int cand) { return (cand ? d.dinv1 : d.dinv) + (long)inst * d.Fpad; }
__device__ __forceinline__ double *mf_vv(const DV &d, int inst, int cand) { return (cand ? d.vv1 : d.vv) + (long)inst * d.Fpad; }
// instance and candidate of a factor-side workgroup; false: nothing to do
__device__ __forceinline__ bool mf_candidate(const DV &d, int want, int &inst, int &cand)
{
    inst = blockIdx.y; cand = 0;
    if (inst >= d.B) { inst -= d.B; cand = 1; }
    if (d.phase[inst] != want) return false;
    return cand == 0 || mf_speculates(d, d.ist[inst]);
}

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

// The same values by the workgroup that has just built the instance's Newton right-hand side (ipm.hip, k_ipm_rhs /
// k_ipm_head; round 4): the flat kernel above was a launch of its own on the critical path of every sweep (~95 us at 512 x
// IEEE-118 for 50 us of gathers, plus its boundary); one workgroup per instance walks its 7 876 destinations in a dozen
// passes whose loads are independent of one another.  Same sums, same order, both candidates.  Large instances (DV::flat)
// keep the flat kernel.
static __device__ __forceinline__ void mf_values_block(const DV &d, int inst, int nthreads)
{
    const MfDev &M = d.mf;
    const IpmState &st = d.ist[inst];
    const bool spec = mf_speculates(d, st);
    const double dw0 = st.dw, dw1 = next_shift(st.dw, st.dw_last), hsc = st.hsc;
    const double *hv = d.hv + (long)inst * d.nnzhc, *jv = d.jv + (long)inst * d.nnzjc;
    const double *Dd = d.Dd + (long)inst * d.m, *sigp = d.sigp + (long)inst * d.n, *hd = d.hd + (long)inst * d.n;
    const int *rt = d.rtype + (long)inst * d.m;
    double *v0 = mf_vals(d, inst, 0), *v1 = spec ? mf_vals(d, inst, 1) : nullptr;
    for (int e = threadIdx.x; e < M.nnzK; e += nthreads) {
        const int k0 = M.item_ptr[e], k1 = M.item_ptr[e + 1];
        double a = 0.0, b = 0.0;
        for (int k = k0; k < k1; ++k) {
            const MfItem it = M.items[k];
            a += mf_item_value(it, hv, jv, Dd, sigp, hd, rt, hsc, dw0);
            if (v1) b += mf_item_value(it, hv, jv, Dd, sigp, hd, rt, hsc, dw1);
        }
        v0[e] = a;
        if (v1) v1[e] = b;
    }
}

// Inertia of a factorisation from its pivot signs (n positive), shared by k_inertia (ipm.hip: one workgroup of the vector
// kernels per instance) and the streamed solve kernel, which does the test of its instance in its own prologue
// (k_mf_solve_top2, mfront.hip: a launch less per sweep).  The counts are small integers: any summation order gives the
// same doubles.
static __device__ __forceinline__ void inertia_count(const DV &d, const double *dinv, const double *dinv1, int nthreads,
                                                     double &np, double &bad, double &np1, double &bad1)
{
    for (int i = threadIdx.x; i < d.Fpad; i += nthreads) {
        if (d.uinv[i] < 0) continue;             // identity padding
        const double v = dinv[i];
        if (!isfinite(v) || v == 0.0) bad += 1; else if (v > 0) np += 1;
        if (dinv1) {
            const double w = dinv1[i];
            if (!isfinite(w) || w == 0.0) bad1 += 1; else if (w > 0) np1 += 1;
        }
    }
}
// ... and the decision (one thread): PH_SOLVE, or a larger delta_w (stays PH_FACTOR).  Sparse path: the sweep has
// factorised the shift st.dw AND -- for the instances mf_speculates() names -- the next shift of the schedule; the
// bookkeeping is that of a run that factorises one shift per sweep (same counters, same decisions as the oracle).
static __device__ __forceinline__ void inertia_decide(const DV &d, int inst, IpmState &st, bool spec, double np, double bad, double np1, double bad1)
{
    const bool ok[2] = { (np == (double)d.n) && bad == 0, spec && (np1 == (double)d.n) && bad1 == 0 };
    for (int cand = 0; cand < (spec ? 2 : 1); ++cand) {
        st.n_factor++;                           // the factorisation with the shift st.dw
        if (ok[cand]) {
            if (st.dw > 0.0) st.dw_last = st.dw;
            st.refine_it = 0; st.sel = cand;
            d.phase[inst] = PH_SOLVE;
            return;
        }
        st.fac_attempt++;
        st.dw = next_shift(st.dw, st.dw_last);   // (candidate 1 was factorised with exactly this shift)
        if (st.dw > 1e40 || st.fac_attempt >= 60) { st.rc = 2; d.phase[inst] = ph_done(d); return; }
    }
}

}  // namespace sqphip
