// ipm.hip -- batched primal-dual interior-point method for the trust-region QP sub-problems.
//
// Seat in the reference: everything behind `JuMP.optimize!(qp.model)` in
// /root/reference/src/algorithms/subproblem_JuMP.jl:178,209,336,388,418 (Ipopt + its linear solver,
// neither vendored).  The mathematical programmes are those of SURVEY.md Appendix A:
//   rows typed as subproblem_JuMP.jl:79-112, shifted as :492-505, trust-region box :432-456,
//   FR slack rule :365-380, L1QP / INFEAS :324-330 / :407-413, LP phase :185-244,
//   results reported as collect_solution! does (:514-563).
//
// Method (DESIGN.md section "IPM"): every row carries elastic variables tp, tm >= 0 so the programme
// always has a strict interior and the start satisfies the (linear) row equations exactly; hard rows
// are priced at rho_big (exact penalty); elastic mass left on a hard row means infeasible (optionally
// confirmed by a phase-1 run, options.ipm_phase1).  Monotone Fiacco-McCormick barrier updates, one Newton direction per iteration from the
// reduced KKT system K = [W J'; J -D] factorised by the batched dense LDL^T (ldlt.hip), inertia
// judged on pivot signs (n positive, m negative) with delta_w escalation, a fixed primal-dual
// regularisation of 1e-8, fraction-to-boundary step lengths.  One 256-thread workgroup owns an instance in the vector kernels; wave-level
// shuffles + a 4-entry LDS exchange do the reductions.
#include "ctx.hpp"
#include "dev_util.hpp"
#include "mf_dev.hpp"
#include <cmath>

#ifndef SQPHIP_VEC_FUSE
#define SQPHIP_VEC_FUSE 1      // rows of H v / J v formed inside the loops that consume them (0: separate passes + barrier)
#endif

namespace sqphip {

// Gathered vectors in LDS (round 3).  The sparse products of the vector stages -- H v, J v, J' w, the expansion of the
// eliminated rows -- walk index -> value chains through global memory: two or three dependent round trips per entry of
// the longest row or column (a bus with ten branches), with one workgroup per instance and one per CU nothing hides
// them.  Where the vectors of an instance fit (DV::vstage doubles of dynamic LDS: n + N), the vector being gathered FROM
// is staged in LDS first -- one coalesced round trip -- and the chain ends in an LDS read.  Same arithmetic, same order.
extern __shared__ double ipm_lds[];

// shader-clock stamps inside the vector stages (instance 0, thread 0; scripts/gpu_mf_trace.py builds with -DSQPHIP_MF_TRACE)
#ifdef SQPHIP_MF_TRACE
__device__ long long g_vec_trace[64];
#define VTR(i) if (blockIdx.x == 0 && threadIdx.x == 0) g_vec_trace[i] = (long long)clock64();
extern "C" int sqphip_vec_trace_read(long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_vec_trace), sizeof(long long) * 64); }
#else
#define VTR(i)
#endif

#define RHO_BIG0 1e4
#define RHO_BIG_MAX 1e10
#define ELASTIC_TOL 1e-8
#define RHO_CERT_FRAC 0.5       // |H p + c| below this fraction of max_j (|J|'|y| + z)_j: the elastic solution certifies infeasibility
// primal-dual regularisation of the Newton matrix (part of the method; residuals are unregularised)
#define IPM_REG_P 1e-8
#define IPM_REG_D 1e-8

// one row of (hsc H + diag hd) v and of J v (the caller has excluded free rows): the sums hess_mul / jac_mul store, for the
// loops that consume a row's product in the thread that formed it (no trip through global memory, no barrier)
__device__ __forceinline__ double hess_row(const DV &d, const double *hv, const double *hd, double hsc, const double *v, int j)
{
    double acc = 0.0;
    const double hdj = hd[j], vj = v[j];
    if (d.hfull) {
        // a completely dense Hessian (every column holds all n rows, ascending): H_jk is read out of COLUMN k at row j -- the
        // mirror of H_kj, the same bits (both triangles come from one COO entry) -- so the threads j, j + 1, ... of a wave read
        // consecutive addresses; same terms in the same order as the walk down column j
        const double *hj = hv + j;
#pragma unroll 8
        for (int k = 0; k < d.n; ++k) acc += hj[(long)k * d.n] * v[k];
        return hsc * acc + hdj * vj;
    }
    const int k0 = d.hcolptr[j], k1 = d.hcolptr[j + 1];
#pragma unroll 4
    for (int k = k0; k < k1; ++k) acc += hv[k] * v[d.hrowval[k]];
    return hsc * acc + hdj * vj;
}
__device__ __forceinline__ double jac_row(const DV &d, const double *jv, const double *v, int i)
{
    double acc = 0.0;
    const int k0 = d.jrowptr[i], k1 = d.jrowptr[i + 1];
#pragma unroll 4
    for (int k = k0; k < k1; ++k) acc += jv[d.jrslot[k]] * v[d.jrcol[k]];
    return acc;
}
// out_j = hsc * (H v)_j + hd_j v_j   (H full symmetric CSC, gather by column)
__device__ void hess_mul(const DV &d, int inst, double hsc, const double *v, double *out)
{
    const double *hv = d.hv + (long)inst * d.nnzhc, *hd = d.hd + (long)inst * d.n;
    for (int j = threadIdx.x; j < d.n; j += TPB) out[j] = hess_row(d, hv, hd, hsc, v, j);
}
// out_i = J_i v over active rows (CSR view)
__device__ void jac_mul(const DV &d, int inst, const double *v, double *out)
{
    const double *jv = d.jv + (long)inst * d.nnzjc;
    const int *rt = d.rtype + (long)inst * d.m;
    for (int i = threadIdx.x; i < d.m; i += TPB) out[i] = rt[i] != ROW_FREE ? jac_row(d, jv, v, i) : 0.0;
}
// (J' w)_j, w from an LDS copy in which the entries of free rows are zero: no row-type gather (the product of a free row
// is then + 0.0 instead of being skipped -- the same sum)
__device__ __forceinline__ double jact_col_masked(const DV &d, const double *jv, const double *w, int j)
{
    double acc = 0.0;
    const int k0 = d.jcolptr[j], k1 = d.jcolptr[j + 1];
#pragma unroll 4
    for (int k = k0; k < k1; ++k) acc += jv[k] * w[d.jrowval[k]];
    return acc;
}
// |z g - mu| of one complementarity pair, the product and the subtraction in one rounding wherever it is taken
__device__ __forceinline__ double compl_err(double z, double g, double mu) { return fabs(fma(z, g, -mu)); }
// (J' w)_j over active rows
__device__ __forceinline__ double jact_col(const DV &d, const double *jv, const int *rt, const double *w, int j)
{
    double acc = 0.0;
    for (int k = d.jcolptr[j]; k < d.jcolptr[j + 1]; ++k) {
        const int i = d.jrowval[k];
        if (rt[i] != ROW_FREE) acc += jv[k] * w[i];
    }
    return acc;
}

// ---------------------------------------------------------------------------------------------
// K2: COO -> CSC with duplicate summation by precomputed gather lists (sqp.jl:94-102, :113-116)
static __device__ void b_qp_gather(const DV &d)
{
    const int inst = blockIdx.x;
    const IpmState &st = d.ist[inst];
    if (!st.start || st.stage != 0) return;
    const double *jc = d.jcoo + (long)inst * d.nnzj_coo, *hc = d.hcoo + (long)inst * d.nnzh_coo;
    double *jv = d.jv + (long)inst * d.nnzjc, *hv = d.hv + (long)inst * d.nnzhc;
    for (int s = threadIdx.x; s < d.nnzjc; s += TPB) {
        double a = 0.0;
        for (int k = d.jg_ptr[s]; k < d.jg_ptr[s + 1]; ++k) a += jc[d.jg_src[k]];
        jv[s] = a;
    }
    for (int s = threadIdx.x; s < d.nnzhc; s += TPB) {
        double a = 0.0;
        for (int k = d.hg_ptr[s]; k < d.hg_ptr[s + 1]; ++k) a += hc[d.hg_src[k]];
        hv[s] = a;
    }
}

__device__ __forceinline__ double push_inside(double v, double lo, double hi)
{
    const double k1 = 1e-2, k2 = 1e-2;
    const bool hl = fin(lo), hu = fin(hi);
    if (hl && hu) {
        const double w = hi - lo;
        const double pl = fmin(k1 * fmax(1.0, fabs(lo)), k2 * w);
        const double pu = fmin(k1 * fmax(1.0, fabs(hi)), k2 * w);
        if (v < lo + pl) v = lo + pl;
        if (v > hi - pu) v = hi - pu;
    } else if (hl) {
        const double pl = k1 * fmax(1.0, fabs(lo));
        if (v < lo + pl) v = lo + pl;
    } else if (hu) {
        const double pu = k1 * fmax(1.0, fabs(hi));
        if (v > hi - pu) v = hi - pu;
    }
    return v;
}

// keep a primal variable a few ulps inside its box (p + a dp may round onto the bound)
__device__ __forceinline__ double nudge_inside(double v, double lo, double hi)
{
    if (fin(lo)) { const double g = 1e-15 * fmax(1.0, fabs(lo)); if (v - lo < g) v = lo + g; }
    if (fin(hi)) { const double g = 1e-15 * fmax(1.0, fabs(hi)); if (hi - v < g) v = hi - g; }
    return v;
}

#define INST_PTRS                                                                                   \
    const long on = (long)inst * d.n, om = (long)inst * d.m;                                       \
    double *c = d.c + on, *hd = d.hd + on, *lb = d.lb + on, *ub = d.ub + on;                        \
    double *lo = d.lo + om, *hi = d.hi + om, *wp = d.wp + om, *wm = d.wm + om;                      \
    int *rt = d.rtype + om, *rb = d.rbase + om, *hard = d.hard + om;                                \
    double *p = d.p + on, *zl = d.zl + on, *zu = d.zu + on;                                         \
    double *s = d.s + om, *tp = d.tp + om, *tm = d.tm + om, *y = d.y + om, *vl = d.vl + om,         \
           *vu = d.vu + om;                                                                         \
    double *dp = d.dp + on, *dzl = d.dzl + on, *dzu = d.dzu + on;                                   \
    double *ds = d.ds + om, *dtp = d.dtp + om, *dtm = d.dtm + om, *dy = d.dy + om,                  \
           *dvl = d.dvl + om, *dvu = d.dvu + om;                                                    \
    double *rd = d.rd + on, *rp = d.rp + om, *sigp = d.sigp + on, *Dd = d.Dd + om;                  \
    double *rhs = d.rhs + (long)inst * d.Npad, *sol = d.sol + (long)inst * d.Npad;                  \
    double *wn = d.wn + on, *wN = d.wN + (long)inst * d.Npad;                                       \
    double *zpv = d.zp + om, *zmv = d.zm + om, *rdir = d.rdir + om;                                 \
    (void)zpv; (void)zmv; (void)rdir;                                                               \
    const double *jv = d.jv + (long)inst * d.nnzjc;                                                 \
    (void)c; (void)hd; (void)lb; (void)ub; (void)lo; (void)hi; (void)wp; (void)wm; (void)rt;        \
    (void)rb; (void)hard; (void)p; (void)zl; (void)zu; (void)s; (void)tp; (void)tm; (void)y;        \
    (void)vl; (void)vu; (void)dp; (void)dzl; (void)dzu; (void)ds; (void)dtp; (void)dtm; (void)dy;   \
    (void)dvl; (void)dvu; (void)rd; (void)rp; (void)sigp; (void)Dd; (void)rhs; (void)sol; (void)wn; \
    (void)wN; (void)jv;

// ---------------------------------------------------------------------------------------------
// Mode set-up (stage 0 only computes the request-dependent data), weights, interior start.
static __device__ void b_ipm_start(const DV &d)
{
    const int inst = blockIdx.x;
    IpmState &st = d.ist[inst];
    if (!st.start) return;
    INST_PTRS
    const int mode = st.mode;
    const double delta = st.delta;
    const double *xk = d.xk + on, *cin = d.cin + on, *bE = d.bE + om;
    const double *xL = d.xL + on, *xU = d.xU + on, *gL = d.gL + om, *gU = d.gU + om;
    const bool use_obj = (mode == SQPHIP_MODE_QP || mode == SQPHIP_MODE_SOC || mode == SQPHIP_MODE_L1QP);
    const bool lp = mode == SQPHIP_MODE_LP;
    // options.ipm_warm_start: the first run of a solve starts from the step and equality multipliers of the previous
    // solved sub-problem of the same mode (still in p / y); restarts with a larger penalty, phase-1 runs start cold
    const bool warm = d.ipm_warm && !lp && st.prev_mode == mode + 1 && st.stage == 0 && st.rho_big == RHO_BIG0;
    const double ysc = warm ? 1.0 / st.sf : 0.0;        // previous objective scale (st.sf is overwritten below)
    // objective scale from the raw gradient
    double cm = 0.0;
    for (int j = threadIdx.x; j < d.n; j += TPB) {
        const double cj = lp ? -2.0 * xk[j] : (use_obj ? cin[j] : 0.0);
        cm = fmax(cm, fabs(cj));
    }
    cm = block_reduce<OpMax>(cm);
    const double sf = cm > 100.0 ? 100.0 / cm : 1.0;
    const double osc = st.stage == 0 ? sf : 0.0;       // phase 1 drops the objective
    for (int j = threadIdx.x; j < d.n; j += TPB) {
        double l, u;
        if (lp) { l = xL[j]; u = xU[j]; }
        else {
            const double vl_ = xL[j] - xk[j], vu_ = xU[j] - xk[j];
            l = fmax(-delta, vl_); u = fmin(delta, vu_);
            if (l > u) { l = fmax(-delta, fmin(0.0, vl_)); u = fmin(delta, fmax(0.0, vu_)); }
        }
        if (fin(l) && fin(u) && u - l < 1e-8) { const double mid = 0.5 * (l + u); l = mid - 5e-9; u = mid + 5e-9; }
        lb[j] = l; ub[j] = u;
        c[j] = osc * (lp ? -2.0 * xk[j] : (use_obj ? cin[j] : 0.0));
        hd[j] = lp ? 2.0 * osc : 0.0;
    }
    const double soft_w = (mode == SQPHIP_MODE_L1QP ? st.mu_pen : 1.0) * sf;
    for (int i = threadIdx.x; i < d.m; i += TPB) {
        const double gl = gL[i], gu = gU[i];
        int ty, hd_;
        if (lp && i >= d.nlin) { ty = ROW_FREE; hd_ = 1; lo[i] = 0.0; hi[i] = 0.0; }
        else {
            if (lp) { lo[i] = gl; hi[i] = gu; } else { lo[i] = gl - bE[i]; hi[i] = gu - bE[i]; }
            if (gl == gu) ty = ROW_EQ;
            else if (gl > -INFINITY || gu < INFINITY) ty = ROW_INEQ;
            else ty = ROW_FREE;
            const bool nonlinear = i >= d.nlin;
            if (mode == SQPHIP_MODE_FR) hd_ = !(nonlinear && !(bE[i] >= gl && bE[i] <= gu));
            else if (mode == SQPHIP_MODE_L1QP || mode == SQPHIP_MODE_INFEAS) hd_ = !nonlinear;
            else hd_ = 1;
        }
        rb[i] = ty; hard[i] = hd_;
        if (st.stage == 1) {      // phase 1: only hard rows, unit weights
            rt[i] = hd_ ? ty : ROW_FREE;
            wp[i] = wm[i] = 1.0;
        } else {
            rt[i] = ty;
            wp[i] = wm[i] = hd_ ? st.rho_big : soft_w;
        }
    }
    __syncthreads();
    // interior start
    const double mu0 = 1.0;
    for (int j = threadIdx.x; j < d.n; j += TPB) {
        const double pj = push_inside(lp ? xk[j] : (warm ? p[j] : 0.0), lb[j], ub[j]);
        p[j] = pj;
        zl[j] = fin(lb[j]) ? mu0 / (pj - lb[j]) : 0.0;
        zu[j] = fin(ub[j]) ? mu0 / (ub[j] - pj) : 0.0;
    }
    __syncthreads();
    jac_mul(d, inst, p, rp);      // rp used as scratch for J p0
    __syncthreads();
    for (int i = threadIdx.x; i < d.m; i += TPB) {
        double si = 0, tpi = 0, tmi = 0, yi = 0, vli = 0, vui = 0;
        if (rt[i] != ROW_FREE) {
            const double v = rp[i];
            si = rt[i] == ROW_EQ ? lo[i] : push_inside(v, lo[i], hi[i]);
            const double dd = si - v;
            if (rt[i] == ROW_INEQ) {
                if (fin(lo[i])) vli = mu0 / (si - lo[i]);
                if (fin(hi[i])) vui = mu0 / (hi[i] - si);
                yi = vli - vui;
                const double cap = 0.5 * fmin(wp[i], wm[i]);
                if (fabs(yi) > cap) { const double sc = cap / fabs(yi); vli *= sc; vui *= sc; yi *= sc; }
            } else if (warm) {       // equality row: the previous multiplier in this solve's objective scale, inside the penalty box
                const double cap = 0.5 * fmin(wp[i], wm[i]);
                yi = fmax(-cap, fmin(cap, y[i] * ysc * sf));
            }
            tpi = fmax(dd, 0.0) + mu0 / (wp[i] - yi);
            tmi = fmax(-dd, 0.0) + mu0 / (wm[i] + yi);
            const double e = (tpi - tmi) - dd;
            if (e > 0) tmi += e; else tpi -= e;
        }
        s[i] = si; tp[i] = tpi; tm[i] = tmi; y[i] = yi; vl[i] = vli; vu[i] = vui;
        zpv[i] = rt[i] == ROW_FREE ? 1.0 : wp[i] - yi; zmv[i] = rt[i] == ROW_FREE ? 1.0 : wm[i] + yi;
    }
    if (threadIdx.x == 0) {
        st.sf = sf; st.soft_w = soft_w; st.hsc = (st.stage == 0 && use_obj) ? sf : 0.0;
        st.mu = 1.0; st.iter = 0; st.rc = -1; st.dw = 0.0; st.dw_floor = 0.0; st.n_acc = 0; st.n_acc2 = 0; st.n_acc3 = 0;
        st.dw_last = 0.0;
        st.cn = 0.0;
        st.mpc = d.ipm_corrector != 0; st.use_soc = 0; st.cavg = 0.0;
        st.start = 0;
        d.phase[inst] = ph_prep(d);
    }
}

// ---------------------------------------------------------------------------------------------
// top of an interior-point iteration: residuals, convergence test, barrier update, diagonals
static __device__ void b_ipm_prepare(const DV &d)
{
    const int inst = blockIdx.x;
    if (d.phase[inst] != ph_prep(d)) return;
    IpmState &st = d.ist[inst];
    INST_PTRS
    // iteration limit; half of it for a second-order correction (sqphip.h, options.ipm_max_iter)
    if (st.iter >= (st.mode == SQPHIP_MODE_SOC ? d.ipm_max_iter / 2 : d.ipm_max_iter)) {
        if (threadIdx.x == 0) { st.rc = 1; d.phase[inst] = ph_done(d); }
        return;
    }
    const double hsc = st.hsc;
    const int n_acc_prev = st.n_acc, n_acc2_prev = st.n_acc2, n_acc3_prev = st.n_acc3;   // read here: thread 0 updates them below, after the reductions' barriers
    VTR(16)
    const double *pv = p, *yv = y;
    if (d.vstage) {                                  // p and y are gathered from below: LDS copies
        double *sp = ipm_lds, *sy = ipm_lds + d.n;
        for (int j = threadIdx.x; j < d.n; j += TPB) sp[j] = p[j];
        for (int i = threadIdx.x; i < d.m; i += TPB) sy[i] = rt[i] != ROW_FREE ? y[i] : 0.0;      // (masked: jact_col_masked)
        __syncthreads();
        pv = sp; yv = sy;
    }
    VTR(17)
    // (H p and J p are formed row by row inside the loops that consume them: SQPHIP_VEC_FUSE=0 restores the two passes
    //  through rd / rp with a barrier behind them -- same sums, same bits)
    const double *hvp = d.hv + (long)inst * d.nnzhc;
    // (d.flat, large instances: H p, J' y and J p were formed by k_sp_products over the whole chip -- the same row / column
    //  sums by the same routines, one thread per row of the batch instead of one workgroup per instance)
    const double *fH = d.flat ? d.fH + on : nullptr, *fJt = d.flat ? d.fJt + on : nullptr, *fJ = d.flat ? d.fJ + om : nullptr;
#if !SQPHIP_VEC_FUSE
    hess_mul(d, inst, hsc, pv, rd);
    VTR(18)
    jac_mul(d, inst, pv, rp);
    __syncthreads();
#endif
    VTR(19)
    double csum = 0, cmax = 0, rdn = 0, rpn = 0, dl1 = 0, nc = 0;
    // ce0: the complementarity error against the barrier value this iteration starts from -- the first pass of the barrier
    // update below, taken in the same loops (the products are the ones summed here) and reduced with the other six
    double ce0 = 0.0;
    const double mu_in = st.mu;
    // (loop bodies: every operand is loaded before the first test on one of them -- a load behind a branch on another
    //  load is a second memory round trip, and these loops are round trips and little else)
    for (int j = threadIdx.x; j < d.n; j += TPB) {
#if SQPHIP_VEC_FUSE
        const double rdj = fH ? fH[j] : hess_row(d, hvp, hd, hsc, pv, j);
#else
        const double rdj = rd[j];
#endif
        const double cj = c[j], pj = p[j], lbj = lb[j], ubj = ub[j], zlj = zl[j], zuj = zu[j];
        double r = rdj + cj - (fJt ? fJt[j] : (d.vstage ? jact_col_masked(d, jv, yv, j) : jact_col(d, jv, rt, yv, j)));
        const double g_l = pj - lbj, g_u = ubj - pj;
        double sg = 0.0;
        if (fin(lbj)) { r -= zlj; const double cc = zlj * g_l; csum += cc; cmax = fmax(cmax, cc); nc += 1; dl1 += zlj; sg += zlj / g_l; ce0 = fmax(ce0, compl_err(zlj, g_l, mu_in)); }
        if (fin(ubj)) { r += zuj; const double cc = zuj * g_u; csum += cc; cmax = fmax(cmax, cc); nc += 1; dl1 += zuj; sg += zuj / g_u; ce0 = fmax(ce0, compl_err(zuj, g_u, mu_in)); }
        rd[j] = r; sigp[j] = sg;
        rdn = fmax(rdn, fabs(r));
    }
    for (int i = threadIdx.x; i < d.m; i += TPB) {
        const int rti = rt[i];
        const double tpi = tp[i], tmi = tm[i], si = s[i], zp = zpv[i], zm = zmv[i], yi = y[i], loi = lo[i], hii = hi[i],
                     vli = vl[i], vui = vu[i];
#if SQPHIP_VEC_FUSE
        const double rpi = fJ ? fJ[i] : (rti != ROW_FREE ? jac_row(d, jv, pv, i) : 0.0);
#else
        const double rpi = rp[i];
#endif
        if (rti == ROW_FREE) { rp[i] = 0.0; Dd[i] = 1.0; continue; }
        const double r = rpi + tpi - tmi - si;
        rp[i] = r; rpn = fmax(rpn, fabs(r));
        double cc = zp * tpi; csum += cc; cmax = fmax(cmax, cc);
        cc = zm * tmi; csum += cc; cmax = fmax(cmax, cc); nc += 2;
        ce0 = fmax(ce0, fmax(compl_err(zp, tpi, mu_in), compl_err(zm, tmi, mu_in)));
        dl1 += fabs(yi);
        double dd = tpi / zp + tmi / zm;
        if (rti == ROW_INEQ) {
            double sig = 0.0;
            if (fin(loi)) { const double al = si - loi; cc = vli * al; csum += cc; cmax = fmax(cmax, cc); nc += 1; sig += vli / al; ce0 = fmax(ce0, compl_err(vli, al, mu_in)); }
            if (fin(hii)) { const double au = hii - si; cc = vui * au; csum += cc; cmax = fmax(cmax, cc); nc += 1; sig += vui / au; ce0 = fmax(ce0, compl_err(vui, au, mu_in)); }
            dd += 1.0 / sig;
        }
        Dd[i] = dd;
    }
    VTR(20)
    block_reduce7<OpSum, OpMax, OpMax, OpMax, OpSum, OpSum, OpMax>(csum, cmax, rdn, rpn, dl1, nc, ce0);
    VTR(21)
    const double cavg = nc > 0 ? csum / nc : 0.0;
    if (!fin(rdn) || !fin(cavg) || !fin(rpn)) {
        if (threadIdx.x == 0) { st.rc = 2; d.phase[inst] = ph_done(d); }
        return;
    }
    const double sd = fmax(100.0, dl1 / (double)(d.n + d.m)) / 100.0;
    const double e0 = fmax(fmax(rdn / sd, rpn), cmax / sd);
    // converged, or acceptable: 8 consecutive iterates within 100 x tol
    const int n_acc = e0 <= 100.0 * d.ipm_tol ? n_acc_prev + 1 : 0;
    const int n_acc2 = e0 <= 1000.0 * d.ipm_tol ? n_acc2_prev + 1 : 0;     // ... or 15 within 1000 x tol
    // ... or 25 within 10^4 x tol: a sub-problem with nearly flat directions stalls there under the regularisation the inertia
    // test demands (9241-bus shape; never fires on the IEEE-118 workload; oracle/qp_ipm.c, ipm_run, has the data)
    const int n_acc3 = e0 <= 1e4 * d.ipm_tol ? n_acc3_prev + 1 : 0;
    if (e0 <= d.ipm_tol || n_acc >= 8 || n_acc2 >= 15 || n_acc3 >= 25) {
        // (which rule ended the run and at what scaled error is reported: sqphip_qp_termination, sqphip_sqp_qp_log_term)
        if (threadIdx.x == 0) { st.rc = 0; st.e0 = e0; st.acc_rule = e0 <= d.ipm_tol ? 0 : (n_acc >= 8 ? 1 : (n_acc2 >= 15 ? 2 : 3)); d.phase[inst] = ph_done(d); }
        return;
    }
    if (threadIdx.x == 0) { st.n_acc = n_acc; st.n_acc2 = n_acc2; st.n_acc3 = n_acc3; }
    // barrier update: mu <- max(mu_min, min(0.2 mu, mu^1.5)) while the barrier problem is solved
    double mu = st.mu;
    const double mu_min = d.ipm_tol / 10.0;
    const int mpc = st.mpc;           // predictor-corrector mode picks mu after the predictor (k_mpc)
    for (int kk = 0; kk < 20 && !mpc; ++kk) {
        double ce = ce0;                 // first pass: taken with the residual loops above (mu is still the value they used)
        if (kk > 0) {
            ce = 0.0;
            for (int j = threadIdx.x; j < d.n; j += TPB) {
                if (fin(lb[j])) ce = fmax(ce, compl_err(zl[j], p[j] - lb[j], mu));
                if (fin(ub[j])) ce = fmax(ce, compl_err(zu[j], ub[j] - p[j], mu));
            }
            for (int i = threadIdx.x; i < d.m; i += TPB) {
                if (rt[i] == ROW_FREE) continue;
                ce = fmax(ce, compl_err(zpv[i], tp[i], mu));
                ce = fmax(ce, compl_err(zmv[i], tm[i], mu));
                if (rt[i] == ROW_INEQ) {
                    if (fin(lo[i])) ce = fmax(ce, compl_err(vl[i], s[i] - lo[i], mu));
                    if (fin(hi[i])) ce = fmax(ce, compl_err(vu[i], hi[i] - s[i], mu));
                }
            }
            ce = block_reduce<OpMax>(ce);
        }
        const double emu = fmax(fmax(rdn / sd, rpn), ce / sd);
        if (emu > 10.0 * mu || mu <= mu_min) break;
        mu = fmax(mu_min, fmin(0.2 * mu, pow(mu, 1.5)));
    }
    if (threadIdx.x == 0) {
        st.mu = mu; st.tau = fmax(0.99, 1.0 - mu); st.e0 = e0; st.cavg = cavg; st.use_soc = 0;
        st.ipm_iters++;
        // st.dw still holds the correction the previous iteration of this solve ended with: if it needed one,
        // skip the zero trial and start from a third of it
        const double keep = st.dw > 3e-10 ? fmax(1e-20, st.dw / 3.0) : 0.0;
        st.dw = keep; st.dw_floor = keep; st.fac_attempt = 0; st.dir_attempt = 0;
        d.phase[inst] = ph_fact(d);
    }
    VTR(22)
}

// ---------------------------------------------------------------------------------------------
// K4: dense KKT assembly, one workgroup per column: zero the column from the diagonal down, then
// scatter H (lower), J and the diagonals.  Padding columns are identity.
// Condensed form (options.kkt_condense, oracle/qp_ipm.c kkt_assemble_condensed): rows with gL != gU have the
// diagonal block -(D + reg) and are eliminated exactly,
//   [ W + J_I' (D_I + reg)^-1 J_I    J_E' ]      order n + mk instead of n + m;
//   [ J_E                       -(D_E + reg) ]
// the inertia rule is unchanged (n positive pivots): the eliminated block is negative definite.
__global__ __launch_bounds__(128) void k_kkt_assemble(DV d)
{
    const int inst = blockIdx.y;
    if (d.phase[inst] != PH_FACTOR) return;
    const int p = blockIdx.x;                    // position (column of the factorised matrix)
    const int u = d.uinv[p];                     // unknown at this position: variable, kept row, or -1 = padding
    const IpmState &st = d.ist[inst];
    double *col = d.K + (long)inst * d.ld * d.Fpad + (long)p * d.ld;
    // zero the column from the diagonal down; in a leading (independent) tile column the tiles between its own
    // diagonal tile and the remainder are never written by anybody -- zeroed once at allocation, they stay zero
    const int lead_end = 64 * d.Ts;
    if (p < lead_end) {
        for (int i = p + threadIdx.x; i < (p | 63) + 1; i += 128) col[i] = 0.0;
        if (d.tmask) {
            // ... and of the panel below only the blocks that couple to this leading tile: the others are never
            // written either (the scatter below puts entries into coupled blocks only, the panel solve skips the rest)
            const int k = p >> 6;
            for (int r = 0; r < (d.Fpad - lead_end) / 64; ++r)
                if (d.tmask[r * d.Ts + k] && threadIdx.x < 64) col[lead_end + 64 * r + threadIdx.x] = 0.0;
        } else {
            for (int i = lead_end + threadIdx.x; i < d.Fpad; i += 128) col[i] = 0.0;
        }
    } else {
        for (int i = p + threadIdx.x; i < d.Fpad; i += 128) col[i] = 0.0;
    }
    __syncthreads();
    const double *jv = d.jv + (long)inst * d.nnzjc;
    const int *rt = d.rtype + (long)inst * d.m;
    const double *Dd = d.Dd + (long)inst * d.m;
    if (u >= 0 && u < d.n && d.hcolptr[u + 1] - d.hcolptr[u] > 256) {
        // a long column (a dense Hessian: round 4): its Hessian entries and the entries of the rows that stay in the matrix
        // go to distinct positions -- every thread of the workgroup scatters its share (one thread walking 1 920 entries per
        // column was 37 % of the GPU time of the dense workload); the diagonal and the eliminated rows, whose contributions
        // meet, stay with thread 0 in the order of the serial walk
        const int j = u;
        const double hsc = d.ist[inst].hsc;
        const double *hv = d.hv + (long)inst * d.nnzhc;
        __shared__ double hdiag;
        if (threadIdx.x == 0) hdiag = 0.0;
        __syncthreads();
        for (int k = d.hcolptr[j] + threadIdx.x; k < d.hcolptr[j + 1]; k += 128) {
            const int i = d.hrowval[k];
            if (i == j) { hdiag = hsc * hv[k]; continue; }
            const int q = d.upos[i];
            if (q > p) col[q] += hsc * hv[k];
        }
        for (int k = d.jcolptr[j] + threadIdx.x; k < d.jcolptr[j + 1]; k += 128) {
            const int i = d.jrowval[k];
            if (rt[i] == ROW_FREE) continue;
            if (!d.condense || d.kpos[i] >= 0) {
                const int q = d.upos[d.n + (d.condense ? d.kpos[i] : i)];
                if (q > p) col[q] += jv[k];
            }
        }
        __syncthreads();
        if (threadIdx.x != 0) return;
        double diag = d.hd[(long)inst * d.n + j] + d.sigp[(long)inst * d.n + j] + d.ist[inst].dw + IPM_REG_P;
        diag += hdiag;
        col[p] = diag;
        if (d.condense)
            for (int k = d.jcolptr[j]; k < d.jcolptr[j + 1]; ++k) {
                const int i = d.jrowval[k];
                if (rt[i] == ROW_FREE || d.kpos[i] >= 0) continue;
                const double f = jv[k] / (Dd[i] + IPM_REG_D);
                for (int t = d.jrowptr[i]; t < d.jrowptr[i + 1]; ++t) {
                    const int q = d.upos[d.jrcol[t]];
                    if (q >= p) col[q] += f * jv[d.jrslot[t]];
                }
            }
        return;
    }
    if (threadIdx.x != 0) return;
    if (u < 0) { col[p] = 1.0; return; }
    if (u >= d.n) {
        // a row of the factorised matrix: diagonal, and its Jacobian entries towards variables placed after it
        const int i = d.condense ? d.krow[u - d.n] : u - d.n;
        col[p] = rt[i] == ROW_FREE ? -1.0 : -(Dd[i] + IPM_REG_D);
        if (rt[i] != ROW_FREE)
            for (int t = d.jrowptr[i]; t < d.jrowptr[i + 1]; ++t) {
                const int q = d.upos[d.jrcol[t]];
                if (q > p) col[q] += jv[d.jrslot[t]];
            }
        return;
    }
    const int j = u;
    const double hsc = st.hsc;
    const double *hv = d.hv + (long)inst * d.nnzhc;
    double diag = d.hd[(long)inst * d.n + j] + d.sigp[(long)inst * d.n + j] + st.dw + IPM_REG_P;
    for (int k = d.hcolptr[j]; k < d.hcolptr[j + 1]; ++k) {
        const int i = d.hrowval[k];
        if (i == j) { diag += hsc * hv[k]; continue; }
        const int q = d.upos[i];
        if (q > p) col[q] += hsc * hv[k];
    }
    col[p] = diag;
    for (int k = d.jcolptr[j]; k < d.jcolptr[j + 1]; ++k) {
        const int i = d.jrowval[k];
        if (rt[i] == ROW_FREE) continue;
        if (!d.condense || d.kpos[i] >= 0) {     // a row that is in the matrix (placed after this variable?)
            const int q = d.upos[d.n + (d.condense ? d.kpos[i] : i)];
            if (q > p) col[q] += jv[k];
            continue;
        }
        // eliminated row i: its share J_i' (D_i + reg)^-1 J_i of this column (entries at or below the diagonal)
        const double f = jv[k] / (Dd[i] + IPM_REG_D);
        for (int t = d.jrowptr[i]; t < d.jrowptr[i + 1]; ++t) {
            const int q = d.upos[d.jrcol[t]];
            if (q >= p) col[q] += f * jv[d.jrslot[t]];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// second-order terms of the predictor (options.ipm_corrector): dz_aff * dx_aff per complementarity pair
#define SOC_PTRS                                                                                    \
    const double *sZL = d.socZL + on, *sZU = d.socZU + on, *sZP = d.socZP + om, *sZM = d.socZM + om, \
                 *sVL = d.socVL + om, *sVU = d.socVU + om;
#define SOCV(a, k) (soc ? (a)[k] : 0.0)

// working vector of the triangular solves from a full-length right-hand side src = [g; b] (published to the
// workgroup by the caller).  Full form: a copy.  Condensed form: g + J_I' (D_I + reg)^-1 b_I on top, b_E below.
__device__ __forceinline__ double solve_vector_at(const DV &d, const double *jv, const double *Dd, const int *rt, const double *src, int p)
{
    {
        const int u = d.uinv[p];
        double v = 0.0;                                           // identity padding
        if (u >= d.n) v = src[d.n + (d.condense ? d.krow[u - d.n] : u - d.n)];
        else if (u >= 0) {
            v = src[u];
            if (d.condense) {
                const int k0 = d.jcolptr[u], k1 = d.jcolptr[u + 1];
#pragma unroll 2
                for (int k = k0; k < k1; ++k) {
                    const int i = d.jrowval[k];
                    const double jk = jv[k];
                    const int rti = rt[i], kp = d.kpos[i];                 // (every operand of the entry before the test)
                    const double si = src[d.n + i], Di = Dd[i];
                    if (rti != ROW_FREE && kp < 0) v += jk * si / (Di + IPM_REG_D);
                }
            }
        }
        return v;
    }
}
__device__ void load_solve_vector(const DV &d, int inst, const double *src, double *xv)
{
    const double *jv = d.jv + (long)inst * d.nnzjc, *Dd = d.Dd + (long)inst * d.m;
    const int *rt = d.rtype + (long)inst * d.m;
    for (int p = threadIdx.x; p < d.Fpad; p += TPB) xv[p] = solve_vector_at(d, jv, Dd, rt, src, p);
}

// Newton right-hand side for centring target tgt (minus the second-order terms when soc), its working copy xv
// for the triangular solves, sol = 0.  Returns max |rhs| (block-wide).
__device__ double build_rhs(const DV &d, int inst, double tgt, bool soc)
{
    INST_PTRS
    SOC_PTRS
    double rn = 0.0;
    for (int j = threadIdx.x; j < d.n; j += TPB) {
        const double rdj = rd[j], lbj = lb[j], ubj = ub[j], pj = p[j], zlj = zl[j], zuj = zu[j];
        const double sl = SOCV(sZL, j), su = SOCV(sZU, j);
        double g = -rdj;
        if (fin(lbj)) { const double gl = pj - lbj; g += (tgt - sl - zlj * gl) / gl; }
        if (fin(ubj)) { const double gu = ubj - pj; g -= (tgt - su - zuj * gu) / gu; }
        rhs[j] = g; rn = fmax(rn, fabs(g));
    }
    for (int i = threadIdx.x; i < d.m; i += TPB) {
        const int rti = rt[i];
        const double zp = zpv[i], zm = zmv[i], tpi = tp[i], tmi = tm[i], rpi = rp[i], si = s[i], loi = lo[i], hii = hi[i],
                     vli = vl[i], vui = vu[i];
        const double sp_ = SOCV(sZP, i), sm_ = SOCV(sZM, i), svl = SOCV(sVL, i), svu = SOCV(sVU, i);
        double b = 0.0;
        if (rti != ROW_FREE) {
            const double cp = tgt - sp_ - zp * tpi, cm = tgt - sm_ - zm * tmi;
            b = -rpi - cp / zp + cm / zm;
            if (rti == ROW_INEQ) {
                double sig = 0.0, t = 0.0;
                if (fin(loi)) { const double al = si - loi; sig += vli / al; t += (tgt - svl - vli * al) / al; }
                if (fin(hii)) { const double au = hii - si; sig += vui / au; t -= (tgt - svu - vui * au) / au; }
                b += t / sig;
            }
        }
        rhs[d.n + i] = b; rn = fmax(rn, fabs(b));
    }
    rn = block_reduce<OpMax>(rn);        // (its barriers also publish rhs to the whole workgroup)
    for (int i = threadIdx.x; i < d.Npad; i += TPB) sol[i] = 0.0;
    if (!d.flat) load_solve_vector(d, inst, rhs, d.xv + (long)inst * d.Fpad);       // (flat: k_sp_load_xv behind this kernel)
    return rn;
}

// before the factorisation: the Newton right-hand side (it does not depend on delta_w) and its working copy
// xv, which the panel kernels of the factorisation turn into L^-1 rhs on the fly (fused forward elimination).
// Runs for every instance in PH_FACTOR, i.e. again before each re-factorisation.  In predictor-corrector mode
// this is the predictor's (affine-scaling, target 0) right-hand side.
static __device__ void b_build_rhs(const DV &d)
{
    const int inst = blockIdx.x;
    if (d.phase[inst] != ph_fact(d)) return;
    IpmState &st = d.ist[inst];
    const double rn = build_rhs(d, inst, st.mpc ? 0.0 : st.mu, false);
    if (threadIdx.x == 0) st.rn = fmax(1.0, rn);
    // ... and the values of the structural entries of the Newton matrix this instance is about to factorise (both shifts of
    // a speculating instance): they depend on what b_ipm_prepare left (D, Sigma) and on delta_w, all known here
    if (d.vals_inline) mf_values_block(d, inst, TPB);
}

// after the factorisation: inertia from pivot signs -> PH_SOLVE, or a larger delta_w (stays PH_FACTOR).
// Sparse path: the sweep has factorised the shift st.dw AND -- for the instances mf_speculates() names -- the next shift
// of the schedule.  The bookkeeping below is that of a run that factorises one shift per sweep (same counters, same
// decisions as the oracle); the second candidate only saves the sweep a failed first shift would have cost.
__global__ __launch_bounds__(TPB, SQPHIP_VEC_WAVES_PER_EU) void k_inertia(DV d)
{
    const int inst = blockIdx.x;
    if (d.phase[inst] != PH_FACTOR) return;
    IpmState &st = d.ist[inst];
    const bool spec = mf_speculates(d, st);
    const double *dinv = d.dinv + (long)inst * d.Fpad, *dinv1 = spec ? d.dinv1 + (long)inst * d.Fpad : nullptr;
    double np = 0, bad = 0, np1 = 0, bad1 = 0;
    inertia_count(d, dinv, dinv1, TPB, np, bad, np1, bad1);
    np = block_reduce<OpSum>(np); bad = block_reduce<OpSum>(bad);
    if (spec) { np1 = block_reduce<OpSum>(np1); bad1 = block_reduce<OpSum>(bad1); }
    if (threadIdx.x == 0) inertia_decide(d, inst, st, spec, np, bad, np1, bad1);
}

// after a triangular solve: accumulate (expanding the eliminated rows in the condensed form), form the residual
// against the full sparse operator, decide.  Condensed form: one refinement step when the residual is above 1e-11
// relative (the elimination puts 1/D-sized terms into the matrix; see oracle/qp_ipm.c, kkt_solve) -- the residual
// becomes the next right-hand side, the sweep runs one more forward/backward solve and calls this kernel with last = 1.
// part: 3 = the whole routine (one workgroup per instance does everything), 1 = accumulation only, 2 = residual and decision
// only (d.flat: the sparse products of each part are formed by flat kernels in front of it, ipm_sweep)
static __device__ void b_refine(const DV &d, int last, int want, int part = 3)
{
    const int inst = blockIdx.x;
    if (d.phase[inst] != want) return;
    IpmState &st = d.ist[inst];
    INST_PTRS
    double *xv = d.xv + (long)inst * d.Fpad;
    const int refine_it = st.refine_it;      // read before the first barrier, written by thread 0 at the end
    const double *fH = d.flat ? d.fH + on : nullptr, *fJt = d.flat ? d.fJt + on : nullptr, *fJ = d.flat ? d.fJ + om : nullptr,
                 *fX = d.flat ? d.fX + om : nullptr;
    VTR(want == PH_RESOLVE ? 0 : 32)
    // LDS copies (d.vstage): xs = the solve's result in variable order, ss = the accumulated solution (both gathered from below)
    double *xs = d.vstage ? ipm_lds : nullptr, *ss = d.vstage ? ipm_lds + d.n : nullptr;
    if (part & 1) {
    if (!d.condense) {
        for (int i = threadIdx.x; i < d.N; i += TPB) {
            const double v = sol[i] + xv[d.upos[i]];
            sol[i] = v;
            if (ss) ss[i] = (i >= d.n && rt[i - d.n] == ROW_FREE) ? 0.0 : v;        // (rows masked: jact_col_masked)
        }
    } else {
        // the right-hand side this solve answered: the Newton rhs, or the residual of the first pass
        const double *cur = refine_it == 0 ? rhs : wN;
        for (int j = threadIdx.x; j < d.n; j += TPB) {
            const double xj = xv[d.upos[j]], v = sol[j] + xj;
            sol[j] = v;
            if (ss) { xs[j] = xj; ss[j] = v; }
        }
        if (xs) __syncthreads();
        for (int i = threadIdx.x; i < d.m; i += TPB) {
            double v;
            const int kp = d.kpos[i], rti = rt[i], t0 = d.jrowptr[i], t1 = d.jrowptr[i + 1];
            const double ci = cur[d.n + i], Di = Dd[i], so = sol[d.n + i];
            if (kp >= 0) v = xv[d.upos[d.n + kp]];
            else if (rti == ROW_FREE) v = -ci;
            else {      // eliminated row: q_i = (J_i dp - b_i) / (D_i + reg)
                double acc = 0.0;
                if (fX) acc = fX[i];
                else if (xs) {
#pragma unroll 4
                    for (int t = t0; t < t1; ++t) acc += jv[d.jrslot[t]] * xs[d.jrcol[t]];
                } else for (int t = t0; t < t1; ++t) acc += jv[d.jrslot[t]] * xv[d.upos[d.jrcol[t]]];
                v = (acc - ci) / (Di + IPM_REG_D);
            }
            const double w = so + v;
            sol[d.n + i] = w;
            if (ss) ss[d.n + i] = rt[i] != ROW_FREE ? w : 0.0;                       // (masked: jact_col_masked)
        }
    }
    }
    if (part == 1) return;
    __syncthreads();
    VTR(want == PH_RESOLVE ? 1 : 33)
    const double hsc = st.hsc;
    const double *sv = ss ? ss : sol;
    // res = rhs - K sol : top block (H + hd + sigp + dw) dp + J' q ; bottom J dp - D q
    const double *hvp = d.hv + (long)inst * d.nnzhc;
#if !SQPHIP_VEC_FUSE
    hess_mul(d, inst, hsc, sv, wn);
    jac_mul(d, inst, sv, wN + d.n);
    __syncthreads();
#endif
    VTR(want == PH_RESOLVE ? 2 : 34)
    double en = 0.0;
    const double dwv = st.dw;
    for (int j = threadIdx.x; j < d.n; j += TPB) {
#if SQPHIP_VEC_FUSE
        const double wnj = fH ? fH[j] : hess_row(d, hvp, hd, hsc, sv, j);
#else
        const double wnj = wn[j];
#endif
        const double sgj = sigp[j], rhj = rhs[j], svj = sv[j];
        const double kx = wnj + (sgj + dwv + IPM_REG_P) * svj +
                          (fJt ? fJt[j] : (ss ? jact_col_masked(d, jv, sv + d.n, j) : jact_col(d, jv, rt, sv + d.n, j)));
        const double r = rhj - kx;
        wN[j] = r; en = fmax(en, fabs(r));
    }
    for (int i = threadIdx.x; i < d.m; i += TPB) {
        const int rti = rt[i];
        const double Ddi = Dd[i], rhi = rhs[d.n + i], soli = sol[d.n + i];
#if SQPHIP_VEC_FUSE
        const double wi = fJ ? fJ[i] : (rti != ROW_FREE ? jac_row(d, jv, sv, i) : 0.0);
#else
        const double wi = wN[d.n + i];
#endif
        const double dd = rti == ROW_FREE ? 1.0 : Ddi + IPM_REG_D;
        const double r = rhi - (wi - dd * soli);
        wN[d.n + i] = r; en = fmax(en, fabs(r));
    }
    en = block_reduce<OpMax>(en);            // (its barriers publish wN)
    VTR(want == PH_RESOLVE ? 3 : 35)
    // no refinement for a predictor (affine-scaling) direction: it only feeds Mehrotra's centring parameter and the
    // second-order terms, the direction that is actually taken -- the corrector -- is refined (oracle: ipm_direction)
    const bool predictor = want == PH_SOLVE && st.mpc;
    const bool stop = last || predictor || refine_it >= 1 || !(en > d.refine_tol * st.rn);
    if (!stop && !d.flat) load_solve_vector(d, inst, wN, xv);
    VTR(want == PH_RESOLVE ? 4 : 36)
    if (threadIdx.x == 0) {
        st.n_solve++;
        st.relres = en / st.rn;
        // predictor-corrector mode: the first solve was the predictor, k_mpc builds the corrector's system
        if (stop) d.phase[inst] = (want == PH_SOLVE && st.mpc) ? PH_MPC : PH_STEP;
        else { st.refine_it++; d.phase[inst] = PH_RESOLVE; st.reload = 1; }  // the residual is the next right-hand side (second solve slot; flat: k_sp_load_xv loads it)
    }
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double ratio(double x, double dx, double a)
{
    if (dx < 0.0) { const double r = -x / dx; if (r < a) a = r; }
    return a;
}

// expand the solution of the reduced system to all directions (centring target tgt, minus the second-order terms
// when soc); ap / ad: this thread's largest primal / dual steps to the boundary
__device__ void expand_directions(const DV &d, int inst, double tgt, bool soc, double &ap, double &ad)
{
    INST_PTRS
    SOC_PTRS
    for (int j = threadIdx.x; j < d.n; j += TPB) {
        const double dpj = sol[j], lbj = lb[j], ubj = ub[j], pj = p[j], zlj = zl[j], zuj = zu[j];
        const double sl = SOCV(sZL, j), su = SOCV(sZU, j);
        dp[j] = dpj;
        double a = 0.0, b = 0.0;
        if (fin(lbj)) { const double gl = pj - lbj; a = (tgt - sl - zlj * gl - zlj * dpj) / gl; ap = ratio(gl, dpj, ap); ad = ratio(zlj, a, ad); }
        if (fin(ubj)) { const double gu = ubj - pj; b = (tgt - su - zuj * gu + zuj * dpj) / gu; ap = ratio(gu, -dpj, ap); ad = ratio(zuj, b, ad); }
        dzl[j] = a; dzu[j] = b;
    }
    for (int i = threadIdx.x; i < d.m; i += TPB) {
        double dyi = 0, dsi = 0, dtpi = 0, dtmi = 0, dvli = 0, dvui = 0;
        const int rti = rt[i];
        const double soli = sol[d.n + i], zp = zpv[i], zm = zmv[i], tpi = tp[i], tmi = tm[i], si = s[i], loi = lo[i], hii = hi[i],
                     vli = vl[i], vui = vu[i];
        const double sp_ = SOCV(sZP, i), sm_ = SOCV(sZM, i), svl = SOCV(sVL, i), svu = SOCV(sVU, i);
        if (rti != ROW_FREE) {
            dyi = -soli;
            dtpi = (tgt - sp_ - zp * tpi + tpi * dyi) / zp;
            dtmi = (tgt - sm_ - zm * tmi - tmi * dyi) / zm;
            ap = ratio(tpi, dtpi, ap); ap = ratio(tmi, dtmi, ap);
            ad = ratio(zp, -dyi, ad); ad = ratio(zm, dyi, ad);
            if (rti == ROW_INEQ) {
                double sig = 0, t = 0, al = 0, au = 0, cl = 0, cu = 0;
                const bool hl = fin(loi), hu = fin(hii);
                if (hl) { al = si - loi; cl = tgt - svl - vli * al; sig += vli / al; t += cl / al; }
                if (hu) { au = hii - si; cu = tgt - svu - vui * au; sig += vui / au; t -= cu / au; }
                dsi = (t - dyi) / sig;
                if (hl) { dvli = (cl - vli * dsi) / al; ap = ratio(al, dsi, ap); ad = ratio(vli, dvli, ad); }
                if (hu) { dvui = (cu + vui * dsi) / au; ap = ratio(au, -dsi, ap); ad = ratio(vui, dvui, ad); }
            }
        }
        dy[i] = dyi; ds[i] = dsi; dtp[i] = dtpi; dtm[i] = dtmi; dvl[i] = dvli; dvu[i] = dvui;
    }
}

// predictor-corrector mode, between the two solves of an iteration (oracle/qp_ipm.c, ipm_run, `if (mpc)`):
// the solution in `sol` is the affine-scaling predictor.  Mehrotra's rule picks the centring parameter from the
// complementarity the predictor would reach, the products dz_aff * dx_aff become second-order terms, and the
// corrector's right-hand side goes through the same factorisation.  If this factorisation needed an inertia
// correction the sub-problem is not convex along the path: the solve falls back to the monotone rule for good,
// restarted from the current average complementarity.
static __device__ void b_mpc(const DV &d)
{
    const int inst = blockIdx.x;
    if (d.phase[inst] != PH_MPC) return;
    IpmState &st = d.ist[inst];
    INST_PTRS
    const double cavg = st.cavg, mu_min = d.ipm_tol / 10.0;
    const bool corr = !(st.dw > 0.0);
    double mu;
    if (corr) {
        double ap = 1e300, ad = 1e300;
        expand_directions(d, inst, 0.0, false, ap, ad);
        block_reduce2<OpMin, OpMin>(ap, ad);
        const double ap1 = fmin(1.0, ap), ad1 = fmin(1.0, ad);
        double *sZL = d.socZL + on, *sZU = d.socZU + on, *sZP = d.socZP + om, *sZM = d.socZM + om,
               *sVL = d.socVL + om, *sVU = d.socVU + om;
        double csum = 0.0, nc = 0.0;
#define PR(zz, dz, xx, dx, dst) do { csum += ((zz) + ad1 * (dz)) * ((xx) + ap1 * (dx)); nc += 1; dst = (dz) * (dx); } while (0)
        for (int j = threadIdx.x; j < d.n; j += TPB) {
            const double lbj = lb[j], ubj = ub[j], pj = p[j], dpj = dp[j], zlj = zl[j], zuj = zu[j], dzlj = dzl[j], dzuj = dzu[j];
            double a = 0.0, b = 0.0;
            if (fin(lbj)) PR(zlj, dzlj, pj - lbj, dpj, a);
            if (fin(ubj)) PR(zuj, dzuj, ubj - pj, -dpj, b);
            sZL[j] = a; sZU[j] = b;
        }
        for (int i = threadIdx.x; i < d.m; i += TPB) {
            const int rti = rt[i];
            const double zp = zpv[i], zm = zmv[i], dyi = dy[i], tpi = tp[i], tmi = tm[i], dtpi = dtp[i], dtmi = dtm[i], si = s[i],
                         dsi = ds[i], loi = lo[i], hii = hi[i], vli = vl[i], vui = vu[i], dvli = dvl[i], dvui = dvu[i];
            double a = 0.0, b = 0.0, e = 0.0, f = 0.0;
            if (rti != ROW_FREE) {
                PR(zp, -dyi, tpi, dtpi, a);
                PR(zm, dyi, tmi, dtmi, b);
                if (rti == ROW_INEQ) {
                    if (fin(loi)) PR(vli, dvli, si - loi, dsi, e);
                    if (fin(hii)) PR(vui, dvui, hii - si, -dsi, f);
                }
            }
            sZP[i] = a; sZM[i] = b; sVL[i] = e; sVU[i] = f;
        }
#undef PR
        block_reduce2<OpSum, OpSum>(csum, nc);
        const double mu_aff = nc > 0 ? csum / nc : 0.0;
        double sigma = cavg > 0.0 ? pow(fmax(0.0, mu_aff) / cavg, 3.0) : 1.0;
        sigma = fmin(1.0, fmax(sigma, 1e-4));
        mu = fmax(mu_min, sigma * cavg);
        __syncthreads();             // second-order terms visible to every thread of build_rhs
    } else {
        mu = fmax(mu_min, fmin(1.0, cavg));
    }
    const double rn = build_rhs(d, inst, mu, corr);
    if (threadIdx.x == 0) {
        st.mu = mu; st.tau = fmax(0.99, 1.0 - mu); st.use_soc = corr ? 1 : 0;
        if (!corr) st.mpc = 0;
        st.rn = fmax(1.0, rn); st.refine_it = 0;
        d.phase[inst] = PH_RESOLVE; st.reload = 2;      // (flat: k_sp_load_xv loads the working vector from rhs)
    }
}

// directions, fraction-to-boundary step lengths, update
static __device__ void b_ipm_step(const DV &d)
{
    const int inst = blockIdx.x;
    if (d.phase[inst] != PH_STEP) return;
    IpmState &st = d.ist[inst];
    INST_PTRS
    double ap = 1e300, ad = 1e300;
    VTR(8)
    expand_directions(d, inst, st.mu, st.use_soc != 0, ap, ad);
    VTR(9)
    block_reduce2<OpMin, OpMin>(ap, ad);
    VTR(10)
    // plain fraction-to-boundary step lengths, primal and dual separately
    const double a = fmin(1.0, st.tau * ap), a_d = fmin(1.0, st.tau * ad);
    const bool ok = fin(st.relres) && st.relres < 1e-6 && fin(a) && fin(a_d);
    if (!ok) {
        if (threadIdx.x == 0) {
            st.dir_attempt++;
            const double fl = st.dw > 0.0 ? 8.0 * st.dw : (st.dw_last > 0.0 ? st.dw_last : 1e-4);
            st.dw_floor = fl; st.dw = fl; st.fac_attempt = 0;
            if (st.dir_attempt >= 12 || fl > 1e20) { st.rc = 2; d.phase[inst] = ph_done(d); }
            else d.phase[inst] = PH_FACTOR;
        }
        return;
    }
    for (int j = threadIdx.x; j < d.n; j += TPB) {
        const double pj = p[j], dpj = dp[j], lbj = lb[j], ubj = ub[j], zlj = zl[j], zuj = zu[j], dzlj = dzl[j], dzuj = dzu[j];
        p[j] = nudge_inside(pj + a * dpj, lbj, ubj); zl[j] = zlj + a_d * dzlj; zu[j] = zuj + a_d * dzuj;
    }
    for (int i = threadIdx.x; i < d.m; i += TPB) {
        const int rti = rt[i];
        const double tpi = tp[i], tmi = tm[i], si = s[i], vli = vl[i], vui = vu[i], yi = y[i], zpi = zpv[i], zmi = zmv[i],
                     dtpi = dtp[i], dtmi = dtm[i], dsi = ds[i], dvli = dvl[i], dvui = dvu[i], dyi = dy[i], loi = lo[i], hii = hi[i];
        if (rti == ROW_FREE) continue;
        tp[i] = tpi + a * dtpi; tm[i] = tmi + a * dtmi;
        double sn = si + a * dsi;
        if (rti == ROW_INEQ) sn = nudge_inside(sn, loi, hii);
        s[i] = sn;
        const double vln = vli + a_d * dvli, vun = vui + a_d * dvui;
        vl[i] = vln; vu[i] = vun;
        y[i] = rti == ROW_INEQ ? vln - vun : yi + a_d * dyi;
        zpv[i] = zpi - a_d * dyi; zmv[i] = zmi + a_d * dyi;
    }
    VTR(11)
    if (threadIdx.x == 0) { st.iter++; d.phase[inst] = ph_prep(d); }
}

// ---------------------------------------------------------------------------------------------
// outcome of a finished interior-point run: accept, phase 1, penalty escalation, or infeasible;
// final results in the JuMP sign convention (collect_solution!, subproblem_JuMP.jl:514-563)
__global__ __launch_bounds__(TPB, SQPHIP_VEC_WAVES_PER_EU) void k_qp_finish(DV d)
{
    const int inst = blockIdx.x;
    if (d.phase[inst] != PH_DONE) return;
    IpmState &st = d.ist[inst];
    INST_PTRS
    double el = 0.0;
    for (int i = threadIdx.x; i < d.m; i += TPB)
        if (rt[i] != ROW_FREE && hard[i]) el = fmax(el, fmax(tp[i], tm[i]));
    el = block_reduce<OpMax>(el);
    int status = -1;       // -1: another run requested
    bool escalate = false;
    if (st.stage == 0) {
        if (st.rc == 1) status = SQPHIP_MOI_ITERATION_LIMIT;
        else if (st.rc == 2) status = SQPHIP_MOI_NUMERICAL_ERROR;
        else if (el <= ELASTIC_TOL) status = SQPHIP_MOI_LOCALLY_SOLVED;
        else if (!d.ipm_phase1) {
            // Elastic mass left on a hard row: infeasible, or is the exact-penalty weight rho too small?  At the
            // solution H p + c = J'y + zl - zu.  For an infeasible programme the right-hand side is a sum of terms of
            // size rho that CANCEL (J'ybar + zbar = 0 is the infeasibility certificate, ybar = y / rho), leaving only
            // the objective gradient; if instead |H p + c| is as large as the terms themselves (no cancellation), the
            // penalty is merely balancing the objective, i.e. rho is too small for this programme: raise it and solve
            // again (oracle/qp_ipm.c has the same rule).  Scale-free: a row with tiny coefficients and a
            // huge multiplier is recognised as such.
            hess_mul(d, inst, st.hsc, p, wn);
            __syncthreads();
            double g = 0.0, a = 0.0;
            for (int j = threadIdx.x; j < d.n; j += TPB) {
                g = fmax(g, fabs(wn[j] + c[j]));
                double t = (fin(lb[j]) ? zl[j] : 0.0) + (fin(ub[j]) ? zu[j] : 0.0);
                for (int k = d.jcolptr[j]; k < d.jcolptr[j + 1]; ++k) {
                    const int i = d.jrowval[k];
                    if (rt[i] != ROW_FREE) t += fabs(jv[k] * y[i]);
                }
                a = fmax(a, t);
            }
            g = block_reduce<OpMax>(g); a = block_reduce<OpMax>(a);
            if (g > RHO_CERT_FRAC * a && st.rho_big < RHO_BIG_MAX) escalate = true;
            else status = SQPHIP_MOI_LOCALLY_INFEASIBLE;
        }
        if (threadIdx.x == 0) st.elastic = el;
    } else {
        const bool infeasible = st.rc != 0 || el > ELASTIC_TOL;
        if (infeasible || st.rho_big >= RHO_BIG_MAX) status = SQPHIP_MOI_LOCALLY_INFEASIBLE;
    }
    __syncthreads();
    if (status < 0) {
        if (threadIdx.x == 0) {
            if (escalate) st.rho_big *= 100.0;
            else if (st.stage == 0) st.stage = 1;
            else { st.stage = 0; st.rho_big *= 100.0; }
            st.start = 1;
            d.phase[inst] = PH_IDLE;
        }
        return;
    }
    double *op = d.op + on, *olam = d.olam + om, *oU = d.omxU + on, *oL = d.omxL + on;
    double *osl = d.oslack + 2 * om;
    if (status == SQPHIP_MOI_LOCALLY_SOLVED) {
        const double sf = st.sf;
        for (int j = threadIdx.x; j < d.n; j += TPB) {
            op[j] = p[j];
            const double rc = ((fin(lb[j]) ? zl[j] : 0.0) - (fin(ub[j]) ? zu[j] : 0.0)) / sf;
            oL[j] = rc > 0 ? rc : 0.0;
            oU[j] = rc < 0 ? rc : 0.0;
        }
        for (int i = threadIdx.x; i < d.m; i += TPB) {
            olam[i] = rt[i] == ROW_FREE ? 0.0 : y[i] / sf;
            osl[i] = tp[i]; osl[d.m + i] = tm[i];
        }
    } else {
        for (int j = threadIdx.x; j < d.n; j += TPB) { op[j] = 0.0; oL[j] = 0.0; oU[j] = 0.0; }
        for (int i = threadIdx.x; i < d.m; i += TPB) { olam[i] = 0.0; osl[i] = 0.0; osl[d.m + i] = 0.0; }
    }
    if (threadIdx.x == 0) { st.status = status; st.prev_mode = status == SQPHIP_MOI_LOCALLY_SOLVED ? st.mode + 1 : 0; d.phase[inst] = PH_IDLE; }
}

__global__ void k_count(DV d)
{
    // counters[0] = instances still iterating, [1] = instances requesting a (re)start
    int run = 0, start = 0;
    for (int i = threadIdx.x; i < d.B; i += blockDim.x) {
        const int ph = d.phase[i];
        if (ph != PH_DONE && ph != PH_IDLE) ++run;
        if (d.ist[i].start) ++start;
    }
    __shared__ int sr[64], ss[64];
    sr[threadIdx.x] = run; ss[threadIdx.x] = start;
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 0, b = 0;
        for (int k = 0; k < (int)blockDim.x; ++k) { a += sr[k]; b += ss[k]; }
        d.counters[0] = a; d.counters[1] = b;
    }
}

// ---------------------------------------------------------------------------------------------
// The vector stages of a sweep as three kernels instead of nine: one workgroup owns one instance in all of them, so
// the stages of an instance simply run one after the other inside its workgroup (a barrier in between publishes the
// phase / state words thread 0 wrote); every stage keeps its own gate.  Same arithmetic as the separate launches.
__global__ __launch_bounds__(TPB) void k_qp_gather(DV d) { b_qp_gather(d); }

// start of a sub-problem (COO -> CSC values, canonical programme), first convergence test, Newton right-hand side
__global__ __launch_bounds__(TPB, SQPHIP_VEC_WAVES_PER_EU) void k_ipm_head(DV d)
{
    b_qp_gather(d);
    __syncthreads();
    b_ipm_start(d);
    __syncthreads();
    b_ipm_prepare(d);
    __syncthreads();
    b_build_rhs(d);
}

// the head of a sweep in which no sub-problem is finished or started (see ipm_sweep): the Newton right-hand side only
__global__ __launch_bounds__(TPB, SQPHIP_VEC_WAVES_PER_EU) void k_ipm_rhs(DV d) { b_build_rhs(d); }

// behind solve slot A: residual check of the first solve, then the corrector's system in predictor-corrector mode
__global__ __launch_bounds__(TPB, SQPHIP_VEC_WAVES_PER_EU) void k_ipm_mid(DV d, int last)
{
    b_refine(d, last, PH_SOLVE);
    if (!d.ipm_corrector) return;
    __syncthreads();
    b_mpc(d);
}

// behind solve slot B: residual check of the second solve, the step, and the convergence test of the new iterate
__global__ __launch_bounds__(TPB, SQPHIP_VEC_WAVES_PER_EU) void k_ipm_tail(DV d, int last)
{
    b_refine(d, last, PH_RESOLVE);
    __syncthreads();
    b_ipm_step(d);
    __syncthreads();
    b_ipm_prepare(d);
}

// Monotone barrier rule (options.ipm_corrector = 0, the default since round 4: what Ipopt, the reference's sub-solver, does
// by default -- mu_strategy = monotone): one solve per factorisation, so everything behind that solve -- residual check,
// step, convergence test of the new iterate, the next Newton right-hand side -- is ONE kernel, and the second solve slot
// of a sweep exists only for the rare refinement step (0.4 % of the iterations on 512 x IEEE-118): `want` = PH_SOLVE behind
// the solve of the sweep, PH_RESOLVE behind a refinement solve.  25 launches per sweep instead of 36.
template <int WANT, bool RHS>
__global__ __launch_bounds__(TPB, SQPHIP_VEC_WAVES_PER_EU) void k_ipm_post(DV d, int last)
{
    b_refine(d, last, WANT);
    __syncthreads();
    b_ipm_step(d);
    __syncthreads();
    b_ipm_prepare(d);
    if (!RHS) return;          // (the right-hand side by k_ipm_rhs behind this kernel: experiment switch SQPHIP_POST_SPLIT)
    __syncthreads();
    b_build_rhs(d);
}

// ---------------------------------------------------------------------------------------------
// Large instances (DV::flat; round 4).  One workgroup per instance is the wrong shape for the vector stages of a few large
// instances (9241-bus shape: 7.8 - 10.2 ms per launch, 38 % of the GPU time at 256 instances, two thirds of it in the sparse
// products: the gathers of one instance go through ONE compute unit's line rate).  The sparse products of a stage -- H v,
// J' w, J v, the expansion of the eliminated rows, the working vector of the solves -- are formed by flat kernels over the
// whole batch (one thread per row or column: thousands of workgroups instead of one per instance), written to buffers, and
// the stage kernels, split at the points where a product depends on what the stage has just computed, read them.  The
// same sums by the same routines in the same order: bit for bit the fused stages (tests force this path on IEEE-118).
enum { SP_PREP = 0, SP_REFINE = 1, SP_XS = 2 };
__global__ __launch_bounds__(256) void k_sp_products(DV d, int what, int want)
{
    const int inst = blockIdx.y;
    if (d.phase[inst] != want) return;
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= d.n + d.m) return;
    const long on = (long)inst * d.n, om = (long)inst * d.m;
    const double *jv = d.jv + (long)inst * d.nnzjc;
    const int *rt = d.rtype + om;
    if (what == SP_XS) {           // J_i x over the solve's result in variable order, for the eliminated rows (b_refine)
        if (t < d.n || !d.condense) return;
        const int i = t - d.n;
        if (d.kpos[i] >= 0 || rt[i] == ROW_FREE) return;
        const double *xv = d.xv + (long)inst * d.Fpad;
        double acc = 0.0;
        for (int k = d.jrowptr[i]; k < d.jrowptr[i + 1]; ++k) acc += jv[d.jrslot[k]] * xv[d.upos[d.jrcol[k]]];
        d.fX[om + i] = acc;
        return;
    }
    // SP_PREP: the iterate (p, y) in front of b_ipm_prepare; SP_REFINE: the accumulated solution in front of the residual
    const double *v = what == SP_PREP ? d.p + on : d.sol + (long)inst * d.Npad;
    const double *w = what == SP_PREP ? d.y + om : v + d.n;
    if (t < d.n) {
        d.fH[on + t] = hess_row(d, d.hv + (long)inst * d.nnzhc, d.hd + on, d.ist[inst].hsc, v, t);
        d.fJt[on + t] = jact_col(d, jv, rt, w, t);
    } else {
        const int i = t - d.n;
        d.fJ[om + i] = rt[i] != ROW_FREE ? jac_row(d, jv, v, i) : 0.0;
    }
}
// working vector of the solves, one thread per position.  mode 0: the Newton right-hand side of every instance about to be
// factorised (behind b_build_rhs); mode 1: instances that asked for it (IpmState.reload: 1 the refinement residual in wN,
// 2 the corrector's right-hand side in rhs)
__global__ __launch_bounds__(256) void k_sp_load_xv(DV d, int mode)
{
    const int inst = blockIdx.y;
    const int rl = d.ist[inst].reload;
    if (mode == 0 ? d.phase[inst] != PH_FACTOR : rl == 0) return;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= d.Fpad) return;
    const double *src = (mode == 0 || rl == 2) ? d.rhs + (long)inst * d.Npad : d.wN + (long)inst * d.Npad;
    d.xv[(long)inst * d.Fpad + p] = solve_vector_at(d, d.jv + (long)inst * d.nnzjc, d.Dd + (long)inst * d.m, d.rtype + (long)inst * d.m, src, p);
}
__global__ void k_sp_clear(DV d)
{
    for (int i = threadIdx.x; i < d.B; i += blockDim.x) d.ist[i].reload = 0;
}
// the stage kernels of a sweep, split where a product has to be formed in between
__global__ __launch_bounds__(TPB, SQPHIP_VEC_WAVES_PER_EU) void k_ipm_head_a(DV d) { b_qp_gather(d); __syncthreads(); b_ipm_start(d); }
__global__ __launch_bounds__(TPB, SQPHIP_VEC_WAVES_PER_EU) void k_ipm_head_b(DV d) { b_ipm_prepare(d); __syncthreads(); b_build_rhs(d); }
__global__ __launch_bounds__(TPB, SQPHIP_VEC_WAVES_PER_EU) void k_ipm_refine_a(DV d, int last, int want) { b_refine(d, last, want, 1); }
__global__ __launch_bounds__(TPB, SQPHIP_VEC_WAVES_PER_EU) void k_ipm_mid_b(DV d, int last)
{
    b_refine(d, last, PH_SOLVE, 2);
    if (!d.ipm_corrector) return;
    __syncthreads();
    b_mpc(d);
}
__global__ __launch_bounds__(TPB, SQPHIP_VEC_WAVES_PER_EU) void k_ipm_tail_b(DV d, int last, int want)
{
    b_refine(d, last, want, 2);
    __syncthreads();
    b_ipm_step(d);
}
__global__ __launch_bounds__(TPB, SQPHIP_VEC_WAVES_PER_EU) void k_ipm_prepare(DV d) { b_ipm_prepare(d); }

void launch_qp_gather(Ctx &C)
{
    hipLaunchKernelGGL(k_qp_gather, dim3(C.d.B), dim3(TPB), 0, C.stream, C.d);
}

static void read_counters(Ctx &C)
{
    hipLaunchKernelGGL(k_count, dim3(1), dim3(64), 0, C.stream, C.d);
    SQPHIP_HIP_OK(hipMemcpyAsync(C.h_counters, C.d.counters, 2 * sizeof(int), hipMemcpyDeviceToHost, C.stream));
    SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
    SQPHIP_HIP_OK(hipGetLastError());       // a failed launch anywhere in the sweep surfaces here, not as a silent wrong answer
}

// One sweep = one pass of the fixed kernel sequence.  Every kernel is gated on per-instance state:
//   k_qp_finish   phase DONE      -> final status (IDLE) or restart request (phase 1 / escalation)
//   [SQP level]   stage gating    -> merit step of finished sub-problems, next sub-problem request
//   k_qp_gather / k_ipm_start     -> instances with a start request
//   k_ipm_prepare phase PREP      -> convergence test, barrier update, diagonals -> FACTOR
//   assemble, rhs, LDL^T (with the forward elimination of the rhs fused in), inertia
//                                 -> FACTOR -> SOLVE (or delta_w bump, stays FACTOR)
//   backward solve, residual check                  -> SOLVE -> STEP
//   k_ipm_step    phase STEP      -> update -> PREP
//   k_ipm_prepare again           -> so that a converged instance is recognised in this sweep
void ipm_sweep(Ctx &C, bool sqp_level)
{
    DV &d = C.d;
    hipStream_t s = C.stream;
    const dim3 gB(d.B), bT(TPB);
    C.n_sweeps++;
    const size_t vlds = 8 * (size_t)d.vstage;
    // Transitions between sub-problems -- k_qp_finish, the stage kernel of run!, the start of the next sub-problem in
    // k_ipm_head -- concern three or four of the 128 instances of a group in any one sweep, and each of these kernels
    // lasts as long as its slowest instance (20 + 95 + 120 us against ~25 us for the right-hand sides of everybody else):
    // with a period P > 1 they run every P-th sweep only, an instance that has finished waits up to P - 1 sweeps for them
    // (gated out of everything meanwhile), and the sweeps in between start with k_ipm_rhs.  Which sweep an instance
    // moves on in changes nothing it computes.  Measured (driver's command, 512 x IEEE-118, groups of 128): P = 1 / 2 / 3 / 4
    // -> 7 443 / 7 682 / 7 769 / 7 763 QP/s with identical work counters; 64 scenarios (groups of 16): 2 020 / 2 023 / 2 002.
    // Default: 3 for groups of 64 instances and more, 2 from 32 on, else 1; SQPHIP_TRANS_PERIOD overrides (read once, at creation).
    // The position is counted per RUN (Ctx::run_sweep, reset by sqp_run_lane): the first sweep of every run is a transition
    // sweep -- a run that starts with every slot idle (scenario queue) draws its scenarios there.
    // (round 4, monotone rule -- shorter sweeps, the transitions weigh more: every fourth sweep from groups of eight on;
    //  512 x IEEE-118 P = 3 / 4 / 5 -> 8 836 / 8 878 / 8 894 QP/s, 64 scenarios P = 1 / 2 / 4 / 5 -> 2 458 / 2 555 / 2 584 / 2 567)
    const int period = C.trans_period > 0 ? C.trans_period : (d.ipm_corrector == 0 ? (d.B >= 8 ? 4 : 1) : (d.B >= 64 ? 3 : (d.B >= 32 ? 2 : 1)));
    const bool trans = !sqp_level || period <= 1 || (C.run_sweep % period) == 0;
    C.run_sweep++;
    const dim3 gP((d.n + d.m + 255) / 256, d.B), gX((d.Fpad + 255) / 256, d.B), b256(256);       // flat products (d.flat)
    // monotone rule: the Newton right-hand side of an iteration is built by the kernel that ends the iteration before
    // (k_ipm_post) or starts the sub-problem (k_ipm_head); the sparse factorisation leaves the working vector of a failed
    // inertia trial as it was, so a sweep without transitions starts with the matrix values (the dense path consumes the
    // vector in place and keeps k_ipm_rhs)
    const bool mono = d.ipm_corrector == 0;
    // Transitions on a side stream (Ctx::side_on; ctx.hpp, the PH_ enum): from the second sweep of a run on, the three
    // transition kernels of EVERY sweep run beside the factorisation / solve / post kernels of the same sweep, on the instances
    // that had finished a sub-problem when the sweep before ended; what they start joins the next sweep.  The main chain loses
    // the transition kernels (every fourth sweep ~330 us for the three or four instances of a group that needed them), an
    // instance waits one sweep instead of 1.5 on average.
    const bool side = sqp_level && C.side_on;
    if (side && C.run_sweep > 1) {           // (run_sweep was incremented above: this is sweep run_sweep - 1 >= 1)
        const long k = C.run_sweep - 1;
        DV ds = d; ds.side = 1;
        SQPHIP_HIP_OK(hipStreamWaitEvent(C.side, C.evC[(k - 1) & 3], 0));
        C.tm.open(C.side);
        hipLaunchKernelGGL(k_qp_finish, gB, bT, 0, C.side, ds);
        sqp_stage_kernels(C, C.side, ds);
        hipLaunchKernelGGL(k_ipm_head, gB, bT, vlds, C.side, ds);
        C.tm.close(KC_TRANS, C.side);
        SQPHIP_HIP_OK(hipEventRecord(C.evS[k & 3], C.side));
    } else if (side || trans) {
        DV d0 = d; d0.side = 0;              // (in line: the first sweep of a run with the side stream, every P-th without)
        C.tm.open(s);
        hipLaunchKernelGGL(k_qp_finish, gB, bT, 0, s, d0);
        if (sqp_level) sqp_stage_kernels(C, s, d0);
        if (!d.flat) hipLaunchKernelGGL(k_ipm_head, gB, bT, vlds, s, d0);
        else {
            hipLaunchKernelGGL(k_ipm_head_a, gB, bT, vlds, s, d);
            hipLaunchKernelGGL(k_sp_products, gP, b256, 0, s, d, (int)SP_PREP, (int)PH_PREP);
            hipLaunchKernelGGL(k_ipm_head_b, gB, bT, vlds, s, d);
        }
        C.tm.close(KC_TRANS, s);
    } else if (!(mono && d.sparse)) hipLaunchKernelGGL(k_ipm_rhs, gB, bT, vlds, s, d);
    if (d.flat && (trans || !mono)) hipLaunchKernelGGL(k_sp_load_xv, gX, b256, 0, s, d, 0);
    if (!d.sparse) hipLaunchKernelGGL(k_kkt_assemble, dim3(d.Fpad, d.B), dim3(128), 0, s, d);
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (C.tm.enabled) { ev = C.tm.get(); hipEventRecord(ev.first, s); }
    // assembly + factorisation, with the forward elimination of xv fused in
    if (d.sparse) { mf_factor(C, PH_FACTOR, true, d.vals_inline != 0); C.tm.n_factor++; }
    else ldlt_factor(C.plan, d.K, d.dinv, d.phase, PH_FACTOR, &C.tm, d.xv, d.vv);
    if (C.tm.enabled) { hipEventRecord(ev.second, s); C.tm.pending_factor.push_back(ev); }
    const bool top_inertia = d.sparse && mf_solve_tests_inertia(C);      // the streamed solve kernel tests the inertia itself
    auto lin_solve = [&](int want, bool skip_fwd) {
        if (d.sparse) mf_solve(C, want, skip_fwd, top_inertia && want == PH_SOLVE);
        else ldlt_solve(C.plan, d.K, d.dinv, d.xv, d.vv, d.phase, want, skip_fwd);
    };
    if (!top_inertia) hipLaunchKernelGGL(k_inertia, gB, bT, 0, s, d);
    if (C.tm.enabled) { ev = C.tm.get(); hipEventRecord(ev.first, s); }
    // backward half of the solve (the forward half happened inside the factorisation), then the residual against
    // the sparse operator.  No iterative refinement: one step of it (the policy until late in round 1, two more
    // launch chains per sweep) changed no iteration count on any test problem -- see oracle/qp_ipm.c, kkt_solve.
    lin_solve(PH_SOLVE, true);
    // Two solve slots per sweep, no host round trip in between.  Slot A (above) is the first solve behind the
    // factorisation; k_refine accumulates it, measures the residual against the sparse operator and sends the instance
    // on: to its step, to k_mpc (a predictor), or -- condensed form, residual above refine_tol -- to a refinement solve.
    // Slot B is one stand-alone forward + backward solve for every instance that has a right-hand side pending
    // (PH_RESOLVE): the corrector of the predictor-corrector mode, or a refinement step.  An instance that needs yet
    // another solve (a refined corrector) keeps PH_RESOLVE and is served by slot B of the next sweep.  (Round 1 read a
    // flag back after each k_refine and launched up to three further solve chains per sweep: two host synchronisations
    // and, with hundreds of instances in flight, 2 forward + 3 backward passes nearly every sweep, each as long as its
    // slowest front chain however few instances took part.)
    const int last = d.condense != 0 ? 0 : 1;          // full form: no refinement
    // (the solve timer brackets the two solve slots separately: the vector stage between them is not a solve kernel)
    if (C.tm.enabled) { hipEventRecord(ev.second, s); C.tm.pending_solve.push_back(ev); }
    auto refine_front = [&](int want) {      // flat: J x of the eliminated rows, the accumulation, the products of the accumulated solution
        hipLaunchKernelGGL(k_sp_products, gP, b256, 0, s, d, (int)SP_XS, want);
        hipLaunchKernelGGL(k_ipm_refine_a, gB, bT, vlds, s, d, last, want);
        hipLaunchKernelGGL(k_sp_products, gP, b256, 0, s, d, (int)SP_REFINE, want);
    };
    auto reload = [&]() {
        hipLaunchKernelGGL(k_sp_load_xv, gX, b256, 0, s, d, 1);
        hipLaunchKernelGGL(k_sp_clear, dim3(1), dim3(256), 0, s, d);
    };
    if (mono) {
        // behind the solve: residual check, step, convergence test, next right-hand side; then -- only when the host has
        // seen an instance ask for it (C.want_resolve: the counter of the sweep before last) -- the refinement solve
        auto post = [&](int want) {
            if (!d.flat) {
                if (C.post_split) {
                    if (want == PH_SOLVE) hipLaunchKernelGGL((k_ipm_post<PH_SOLVE, false>), gB, bT, vlds, s, d, last);
                    else hipLaunchKernelGGL((k_ipm_post<PH_RESOLVE, false>), gB, bT, vlds, s, d, last);
                    hipLaunchKernelGGL(k_ipm_rhs, gB, bT, vlds, s, d);
                } else if (want == PH_SOLVE) hipLaunchKernelGGL((k_ipm_post<PH_SOLVE, true>), gB, bT, vlds, s, d, last);
                else hipLaunchKernelGGL((k_ipm_post<PH_RESOLVE, true>), gB, bT, vlds, s, d, last);
                return;
            }
            refine_front(want);
            hipLaunchKernelGGL(k_ipm_tail_b, gB, bT, vlds, s, d, last, want);
            reload();
            hipLaunchKernelGGL(k_sp_products, gP, b256, 0, s, d, (int)SP_PREP, (int)PH_PREP);
            hipLaunchKernelGGL(k_ipm_head_b, gB, bT, vlds, s, d);
            hipLaunchKernelGGL(k_sp_load_xv, gX, b256, 0, s, d, 0);
        };
        C.tm.open(s);
        post(PH_SOLVE);
        C.tm.close(KC_POST, s);
        // (the refinement slot at most every fourth sweep of a run: with 128 instances in a group one of them asks for it in a
        //  third of all sweeps, and the slot costs every instance of the group a forward and a backward pass of gated kernels;
        //  the few that wait are 0.4 % of the iterations)
        if (C.want_resolve && (!sqp_level || d.B < 8 || (C.run_sweep & 3) == 0)) {
            if (C.tm.enabled) { ev = C.tm.get(); hipEventRecord(ev.first, s); }
            lin_solve(PH_RESOLVE, false);
            if (C.tm.enabled) { hipEventRecord(ev.second, s); C.tm.pending_solve.push_back(ev); }
            post(PH_RESOLVE);
        }
        return;
    }
    if (!d.flat) hipLaunchKernelGGL(k_ipm_mid, gB, bT, vlds, s, d, last);
    else { refine_front(PH_SOLVE); hipLaunchKernelGGL(k_ipm_mid_b, gB, bT, vlds, s, d, last); reload(); }
    if (C.tm.enabled) { ev = C.tm.get(); hipEventRecord(ev.first, s); }
    lin_solve(PH_RESOLVE, false);
    if (C.tm.enabled) { hipEventRecord(ev.second, s); C.tm.pending_solve.push_back(ev); }
    if (!d.flat) hipLaunchKernelGGL(k_ipm_tail, gB, bT, vlds, s, d, last);
    else {
        refine_front(PH_RESOLVE);
        hipLaunchKernelGGL(k_ipm_tail_b, gB, bT, vlds, s, d, last, (int)PH_RESOLVE);
        reload();
        hipLaunchKernelGGL(k_sp_products, gP, b256, 0, s, d, (int)SP_PREP, (int)PH_PREP);
        hipLaunchKernelGGL(k_ipm_prepare, gB, bT, vlds, s, d);
    }
}

// Runs every instance whose IpmState.start flag is set until each has a final MOI status
// (drop-in sqphip_qp_solve path: no SQP-level kernels).
void ipm_run_all(Ctx &C)
{
    C.want_resolve = true;               // (the drop-in seat: every sweep carries the refinement slot)
    for (long sweep = 0; sweep < 100000000L; ++sweep) {
        ipm_sweep(C, false);
        // counters[0] = instances iterating, [1] = start requests (phase 1 / escalation restarts)
        hipLaunchKernelGGL(k_qp_finish, dim3(C.d.B), dim3(TPB), 0, C.stream, C.d);
        read_counters(C);
        if (C.tm.pending_trailing.size() > 4096) C.tm.flush();
        if (C.h_counters[0] == 0 && C.h_counters[1] == 0) break;
    }
}

}  // namespace sqphip
