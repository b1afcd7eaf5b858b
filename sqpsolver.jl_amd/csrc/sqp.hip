// sqp.hip -- device-resident SQP-TR outer loop for a batch of instances, and the merit kernels.
//
// Restates, per instance and entirely in HBM, what `run!(sqp::AbstractSqpTrOptimizer)` does
// (/root/reference/src/algorithms/sqp_trust_region.jl:98-223) with the state of
// sqp_trust_region.jl:26-91, plus
//   violation_of_linear_constraints :237-254,  sub_optimize_lp! :264-304 (dropzeros! utils.jl:16-22),
//   compute_step! :370-380,  sub_optimize_soc! :341-360,  compute_qmodel :487-508,  do_step! :515-579,
//   eval_functions! sqp.jl:86-104,  compute_phi sqp.jl:170-183,  terminate_by_iterlimit sqp.jl:215-224,
//   norm_violations / KT_residuals common.jl:54-77 / :14-23.
// The quirks of the reference are reproduced; `literal_quirks = 0` switches
// #2/#3 to the textbook signs.  The callbacks are the device ACOPF evaluator (acopf_dev.hpp).
// One 256-thread workgroup per instance; the host only sequences the kernels and the sub-solves.
#include "ctx.hpp"
#include "dev_util.hpp"
#include "acopf_dev.hpp"
#include <cmath>
#include <chrono>
#include <thread>

namespace sqphip {

// where an instance stands in run! (sqp_trust_region.jl:124-214); kernels act only on their stage
enum { ST_TOP = 0, ST_QP = 1, ST_SOC = 2, ST_LP = 3, ST_DONE = 4 };

// the sub-problem requested by this instance has reached a final MOI status
static __device__ __forceinline__ bool qp_final(const DV &d, int inst)
{
    const IpmState &I = d.ist[inst];
    return d.phase[inst] == PH_IDLE && I.start == 0 && I.status > 0;
}

#define SQP_PTRS                                                                                     \
    const long on = (long)inst * d.n, om = (long)inst * d.m;                                        \
    SqpState &S = d.sst[inst];                                                                       \
    IpmState &I = d.ist[inst];                                                                       \
    double *x = d.x + on, *lam = d.lambda + om, *mxL = d.mxL + on, *mxU = d.mxU + on;                \
    double *df = d.df + on, *E = d.E + om, *ps = d.pstep + on, *psoc = d.psoc + on;                  \
    double *plam = d.plam + om, *pmxL = d.pmxL + on, *pmxU = d.pmxU + on, *Esoc = d.Esoc + om;       \
    double *tmpx = d.tmpx + on, *tmpE = d.tmpE + om, *hlam = d.hlam + om;                            \
    const double *xL = d.xL + on, *xU = d.xU + on, *gL = d.gL + om, *gU = d.gU + om;                 \
    double *jcoo = d.jcoo + (long)inst * d.nnzj_coo, *hcoo = d.hcoo + (long)inst * d.nnzh_coo;       \
    double *jv = d.jv + (long)inst * d.nnzjc, *hv = d.hv + (long)inst * d.nnzhc;                     \
    (void)x; (void)lam; (void)mxL; (void)mxU; (void)df; (void)E; (void)ps; (void)psoc; (void)plam;   \
    (void)pmxL; (void)pmxU; (void)Esoc; (void)tmpx; (void)tmpE; (void)hlam; (void)xL; (void)xU;      \
    (void)gL; (void)gU; (void)jcoo; (void)hcoo; (void)jv; (void)hv; (void)I; (void)S;

// Julia isapprox(a,b): rtol = sqrt(eps), atol = 0 (sqp_trust_region.jl:146,:200,:535)
static __device__ __forceinline__ bool isapprox_d(double a, double b)
{
    if (a == b) return true;
    if (!fin(a) || !fin(b)) return false;
    return fabs(a - b) <= 1.4901161193847656e-08 * fmax(fabs(a), fabs(b));
}

// common.jl:54-77 with p = 1
static __device__ __forceinline__ double viol1(const DV &d, const double *E, const double *gL, const double *gU,
                               const double *x, const double *xL, const double *xU)
{
    double acc = 0.0;
    for (int i = threadIdx.x; i < d.m; i += TPB) {
        if (E[i] > gU[i]) acc += E[i] - gU[i];
        else if (E[i] < gL[i]) acc += gL[i] - E[i];
    }
    for (int j = threadIdx.x; j < d.n; j += TPB) {
        if (x[j] > xU[j]) acc += x[j] - xU[j];
        else if (x[j] < xL[j]) acc += xL[j] - x[j];
    }
    return block_reduce<OpSum>(acc);
}

static __device__ __forceinline__ double norm_inf(const double *v, int k)
{
    double a = 0.0;
    for (int i = threadIdx.x; i < k; i += TPB) a = fmax(a, fabs(v[i]));
    return block_reduce<OpMax>(a);
}

static __device__ __forceinline__ void gather_csc(const DV &d, const double *jcoo, const double *hcoo, double *jv, double *hv)
{
    for (int s = threadIdx.x; s < d.nnzjc; s += TPB) {
        double a = 0.0;
        for (int k = d.jg_ptr[s]; k < d.jg_ptr[s + 1]; ++k) a += jcoo[d.jg_src[k]];
        jv[s] = a;
    }
    if (hv)
        for (int s = threadIdx.x; s < d.nnzhc; s += TPB) {
            double a = 0.0;
            for (int k = d.hg_ptr[s]; k < d.hg_ptr[s + 1]; ++k) a += hcoo[d.hg_src[k]];
            hv[s] = a;
        }
}

// common.jl:14-23 on the CSC Jacobian; sgn = +1 literal, -1 textbook (lambda and mult_x_U negated)
static __device__ __forceinline__ double kt_residuals(const DV &d, const double *df, const double *lam, const double *mxU,
                                      const double *mxL, const double *jv, double sgn, double *rowsq)
{
    for (int i = threadIdx.x; i < d.m; i += TPB) {
        double a = 0.0;
        for (int k = d.jrowptr[i]; k < d.jrowptr[i + 1]; ++k) { const double v = jv[d.jrslot[k]]; a += v * v; }
        rowsq[i] = a;
    }
    double res = 0.0, sc = 1.0;
    for (int j = threadIdx.x; j < d.n; j += TPB) {
        double jtl = 0.0;
        for (int k = d.jcolptr[j]; k < d.jcolptr[j + 1]; ++k) jtl += jv[k] * lam[d.jrowval[k]];
        res = fmax(res, fabs(df[j] + sgn * jtl + sgn * mxU[j] - mxL[j]));
        sc = fmax(sc, fmax(fabs(df[j]), fmax(fabs(mxU[j]), fabs(mxL[j]))));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < d.m; i += TPB) sc = fmax(sc, fabs(lam[i]) * sqrt(rowsq[i]));
    res = block_reduce<OpMax>(res);
    sc = block_reduce<OpMax>(sc);
    return res / sc;
}

static __device__ __forceinline__ void push_trace(const DV &d, int inst, SqpState &S, double pn)
{
    if (threadIdx.x == 0) {
        if (S.trace_len < SQPHIP_TRACE_CAP) {
            double *r = d.trace + ((long)inst * SQPHIP_TRACE_CAP + S.trace_len) * SQPHIP_TRACE_COLS;
            r[0] = S.iter; r[1] = S.step_acceptance; r[2] = S.fr; r[3] = S.sub_status; r[4] = S.it_ipm;
            r[5] = S.f; r[6] = S.phi; r[7] = S.mu; r[8] = S.Delta; r[9] = pn; r[10] = S.prim_infeas;
            r[11] = S.dual_infeas;
        }
        S.trace_len++;
    }
}

// work of a finished sub-problem, booked under its mode (sqphip_get_mode_counters)
static __device__ __forceinline__ void book_mode(SqpState &S, const IpmState &I)
{
    const int k = I.mode & 3;
    S.md_qp[k]++; S.md_ipm[k] += I.ipm_iters; S.md_fac[k] += I.n_factor;
    int *q = S.qlog + 4 * (S.qlog_n % SQPHIP_QLOG_CAP);
    q[0] = I.mode; q[1] = I.status; q[2] = I.ipm_iters; q[3] = I.n_factor;
    S.qerr[S.qlog_n % SQPHIP_QLOG_CAP] = (float)I.e0; S.qrule[S.qlog_n % SQPHIP_QLOG_CAP] = (signed char)(I.rc == 0 ? I.acc_rule : -1);
    if (I.rc == 0) S.term_rule[I.acc_rule & 3]++;
    S.qlog_n++;
}

static __device__ __forceinline__ void qp_request(IpmState &I, int mode, double delta, double mu_pen)
{
    I.mode = mode; I.delta = delta; I.mu_pen = mu_pen;
    I.stage = 0; I.rho_big = 1e4; I.start = 1; I.ipm_iters = 0; I.n_factor = 0; I.n_solve = 0; I.status = 0;
}

// sqp_trust_region.jl:215-222
static __device__ __forceinline__ void finalize(const DV &d, int inst, SqpState &S, const double *x)
{
    double f;
    __shared__ double fsh;
    acopf_eval(d, inst, x, 1.0, nullptr, &fsh, nullptr, nullptr, nullptr, nullptr);
    __syncthreads();
    f = fsh;
    if (threadIdx.x == 0) { S.obj_val = f; S.done = 1; S.stage = ST_DONE; }
}

// ---------------------------------------------------------------------------------------------
// state of a run about to start from x0; keep_totals: the cumulative work counters survive (a slot of the scenario queue)
static __device__ __forceinline__ void reset_instance(const DV &d, int inst, bool keep_totals)
{
    SQP_PTRS
    const double *x0 = d.x0 + on;
    for (int j = threadIdx.x; j < d.n; j += TPB) { x[j] = x0[j]; mxL[j] = 0; mxU[j] = 0; ps[j] = 0; psoc[j] = 0; }
    for (int i = threadIdx.x; i < d.m; i += TPB) { lam[i] = 0; E[i] = 0; }
    if (threadIdx.x == 0) {
        SqpState z = {};
        z.phi = 1e20; z.mu = d.init_mu; z.Delta = d.tr_size;
        z.prim_infeas = INFINITY; z.dual_infeas = INFINITY;
        z.step_acceptance = 1; z.fr = 0; z.iter = 1; z.ret = -5;
        if (keep_totals) {
            z.n_qp = S.n_qp; z.tot_ipm = S.tot_ipm; z.tot_fac = S.tot_fac; z.tot_sol = S.tot_sol; z.budget = S.budget;
            for (int k = 0; k < 4; ++k) { z.md_qp[k] = S.md_qp[k]; z.md_ipm[k] = S.md_ipm[k]; z.md_fac[k] = S.md_fac[k]; z.term_rule[k] = S.term_rule[k]; }
        }
        S = z;
        I.start = 0; I.dw_last = 0.0; I.prev_mode = 0;
        d.phase[inst] = PH_IDLE;
    }
}

__global__ __launch_bounds__(TPB) void k_sqp_reset(DV d) { reset_instance(d, blockIdx.x, false); }

// run! prologue: sqp_trust_region.jl:100-122
static __device__ __forceinline__ void b_sqp_begin(const DV &d)
{
    const int inst = blockIdx.x;
    SQP_PTRS
    const bool go = !(S.started || S.done);
    __syncthreads();                 // every thread has read the gate before any thread changes the state
    if (!go) return;
    __shared__ double fsh;
    acopf_eval(d, inst, x, 1.0, nullptr, &fsh, nullptr, E, nullptr, nullptr);
    __syncthreads();
    const double f = fsh;
    double lpv = 0.0;                                     // :244-253
    for (int i = threadIdx.x; i < d.nlin; i += TPB) { lpv += fmax(0.0, gL[i] - E[i]); lpv -= fmin(0.0, gU[i] - E[i]); }
    for (int j = threadIdx.x; j < d.n; j += TPB) { lpv += fmax(0.0, xL[j] - x[j]); lpv -= fmin(0.0, xU[j] - x[j]); }
    lpv = block_reduce<OpSum>(lpv);
    if (threadIdx.x == 0) { S.f = f; S.started = 1; S.it_ipm = 0; S.stage = ST_TOP; }
    if (isnan(f)) {                                       // :113-115
        if (threadIdx.x == 0) { S.ret = -13; S.done = 1; S.stage = ST_DONE; }
        return;
    }
    if (lpv > d.tol_infeas) {                             // :116-119 -> sub_optimize_lp! :264-304
        acopf_eval(d, inst, x, 1.0, nullptr, nullptr, df, nullptr, jcoo, nullptr);
        double *xk = d.xk + on;
        for (int j = threadIdx.x; j < d.n; j += TPB) xk[j] = x[j];
        if (threadIdx.x == 0) { qp_request(I, SQPHIP_MODE_LP, S.Delta, S.mu); S.stage = ST_LP; }
    }
}

__global__ __launch_bounds__(TPB) void k_sqp_begin(DV d) { b_sqp_begin(d); }

static __device__ __forceinline__ void b_sqp_lp_finish(const DV &d)
{
    const int inst = blockIdx.x;
    SQP_PTRS
    const bool go = !(S.done || S.stage != ST_LP || !qp_final(d, inst));
    __syncthreads();                 // gate read by every thread before thread 0 moves the stage on
    if (!go) return;
    const double *op = d.op + on, *ol = d.olam + om, *oU = d.omxU + on, *oL = d.omxL + on;
    auto dz = [](double v) { return fabs(v) < 1e-10 ? 0.0 : v; };   // utils.jl:16-22
    for (int j = threadIdx.x; j < d.n; j += TPB) { x[j] = dz(op[j]); mxU[j] = dz(oU[j]); mxL[j] = dz(oL[j]); }
    for (int i = threadIdx.x; i < d.m; i += TPB) lam[i] = dz(ol[i]);
    if (threadIdx.x == 0) {
        S.sub_status = I.status; S.stage = ST_TOP; S.n_qp++; S.it_ipm = I.ipm_iters;
        S.tot_ipm += I.ipm_iters; S.tot_fac += I.n_factor; S.tot_sol += I.n_solve;
        book_mode(S, I);
    }
    __syncthreads();
    push_trace(d, inst, S, norm_inf(ps, d.n));            // print(sqp, "LP")
}

// top of the loop: iteration limit, eval_functions!, infeasibility measures, QP request
// (sqp_trust_region.jl:126-141)
static __device__ __forceinline__ void b_sqp_top(const DV &d)
{
    const int inst = blockIdx.x;
    SQP_PTRS
    const bool go = !(S.done || !S.started || S.stage != ST_TOP || S.budget <= 0);
    const int step_acceptance = S.step_acceptance, fr = S.fr;
    __syncthreads();                 // gate (and the flags used below) read by every thread before any write
    if (!go) return;
    if (threadIdx.x == 0) { S.stage = ST_QP; S.it_ipm = 0; }
    __syncthreads();
    if (S.iter > d.max_iter) {                            // sqp.jl:215-224
        if (threadIdx.x == 0) S.ret = S.prim_infeas <= d.tol_infeas ? 6 : -1;
        __syncthreads();
        finalize(d, inst, S, x);
        return;
    }
    if (step_acceptance) {                                // :134-138, sqp.jl:86-104
        const double hs = d.literal_quirks ? 1.0 : -1.0;
        for (int i = threadIdx.x; i < d.m; i += TPB) hlam[i] = hs * lam[i];
        __syncthreads();
        __shared__ double fsh;
        acopf_eval(d, inst, x, 1.0, hlam, &fsh, df, E, jcoo, d.nnzh_coo ? hcoo : nullptr);
        __syncthreads();
        gather_csc(d, jcoo, hcoo, jv, d.nnzh_coo ? hv : nullptr);
        __syncthreads();
        const double pr = viol1(d, E, gL, gU, x, xL, xU);
        const double du = kt_residuals(d, df, lam, mxU, mxL, jv, hs, tmpE);
        if (threadIdx.x == 0) { S.f = fsh; S.prim_infeas = pr; S.dual_infeas = du; }
    }
    // QP request: QpData(sqp) sqp.jl:66-79, dispatch :314-331
    double *xk = d.xk + on, *cin = d.cin + on, *bE = d.bE + om;
    for (int j = threadIdx.x; j < d.n; j += TPB) { xk[j] = x[j]; cin[j] = df[j]; }
    for (int i = threadIdx.x; i < d.m; i += TPB) bE[i] = E[i];
    if (threadIdx.x == 0) qp_request(I, fr ? SQPHIP_MODE_FR : SQPHIP_MODE_QP, S.Delta, S.mu);
}

// q(p) of sqp_trust_region.jl:487-508 (with_step = true); tmpx/tmpE are scratch
static __device__ __forceinline__ double qmodel_step(const DV &d, int inst, const SqpState &S, const double *p,
                                     const double *x, const double *df, const double *E, const double *jv,
                                     const double *hv, const double *gL, const double *gU, const double *xL,
                                     const double *xU, double *tmpx, double *tmpE)
{
    double acc = 0.0;
    for (int j = threadIdx.x; j < d.n; j += TPB) {
        double hp = 0.0;
        if (d.hfull) { const double *hj = hv + j; for (int k = 0; k < d.n; ++k) hp += hj[(long)k * d.n] * p[k]; }      // (dense Hessian: the mirrored entries, coalesced: ipm.hip hess_row)
        else for (int k = d.hcolptr[j]; k < d.hcolptr[j + 1]; ++k) hp += hv[k] * p[d.hrowval[k]];
        acc += df[j] * p[j] + 0.5 * p[j] * hp;
        tmpx[j] = x[j] + p[j];
    }
    for (int i = threadIdx.x; i < d.m; i += TPB) {
        double jp = 0.0;
        for (int k = d.jrowptr[i]; k < d.jrowptr[i + 1]; ++k) jp += jv[d.jrslot[k]] * p[d.jrcol[k]];
        tmpE[i] = E[i] + jp;
    }
    acc = block_reduce<OpSum>(acc);
    __syncthreads();
    return acc + S.mu * viol1(d, tmpE, gL, gU, tmpx, xL, xU);
}

static __device__ __forceinline__ void accept_step(const DV &d, double *x, double *lam, double *mxL, double *mxU,
                                   const double *step, const double *plam, const double *pmxL,
                                   const double *pmxU)
{
    for (int j = threadIdx.x; j < d.n; j += TPB) { x[j] += step[j]; mxL[j] += pmxL[j]; mxU[j] += pmxU[j]; }
    for (int i = threadIdx.x; i < d.m; i += TPB) lam[i] += plam[i];
}

// after the QP: compute_step!, status branches, phi, termination tests, do_step!
// (sqp_trust_region.jl:141-213, :370-380, :515-579)
static __device__ __forceinline__ void b_sqp_mid(const DV &d)
{
    const int inst = blockIdx.x;
    SQP_PTRS
    const bool go = !(S.done || S.stage != ST_QP || !qp_final(d, inst));
    // snapshot of the flags the branches below test: thread 0 changes S.fr / S.step_acceptance inside those
    // branches, and a wave that reads them late must not take a different path from the one that wrote them
    const int fr = S.fr, step_acceptance = S.step_acceptance;
    __syncthreads();
    if (!go) return;
    const double *op = d.op + on, *ol = d.olam + om, *oU = d.omxU + on, *oL = d.omxL + on;
    // compute_step! :373-378
    for (int j = threadIdx.x; j < d.n; j += TPB) { ps[j] = op[j]; pmxL[j] = oL[j] - mxL[j]; pmxU[j] = oU[j] - mxU[j]; }
    for (int i = threadIdx.x; i < d.m; i += TPB) plam[i] = ol[i] - lam[i];
    __syncthreads();
    const double nl_ = norm_inf(lam, d.m), nL = norm_inf(mxL, d.n), nU = norm_inf(mxU, d.n);
    const double pn = norm_inf(ps, d.n);
    const int st = I.status;
    if (threadIdx.x == 0) {
        S.mu = fmax(fmax(S.mu, nl_), fmax(nL, nU));
        S.sub_status = st; S.n_qp++; S.it_ipm += I.ipm_iters;
        S.tot_ipm += I.ipm_iters; S.tot_fac += I.n_factor; S.tot_sol += I.n_solve;
        book_mode(S, I);
    }
    __syncthreads();
    if (st == SQPHIP_MOI_LOCALLY_SOLVED) {
        if (S.Delta == 1e8 && isapprox_d(pn, S.Delta)) {            // :146-150
            if (threadIdx.x == 0) S.ret = 4;
            __syncthreads();
            finalize(d, inst, S, x);
            return;
        }
    } else if (st == SQPHIP_MOI_LOCALLY_INFEASIBLE) {
        if (fr) {                                                      // :152-159
            if (threadIdx.x == 0) S.ret = S.prim_infeas <= d.tol_infeas ? 6 : 2;
            __syncthreads();
            finalize(d, inst, S, x);
        } else {                                                       // :160-168
            if (threadIdx.x == 0) S.fr = 1;
            __syncthreads();
            push_trace(d, inst, S, pn);
            if (threadIdx.x == 0) { S.iter += 1; S.stage = ST_TOP; S.budget -= 1; }
        }
        return;
    } else {                                                           // :169-178 (quirk #1)
        if (threadIdx.x == 0 && S.prim_infeas <= d.tol_infeas * 10.0) S.ret = 6;
        __syncthreads();
        finalize(d, inst, S, x);
        return;
    }
    if (step_acceptance) {                                             // :180-182, sqp.jl:170-183 alpha = 0
        const double v = viol1(d, E, gL, gU, x, xL, xU);
        if (threadIdx.x == 0) S.phi = fr ? v : S.f + S.mu * v;
    }
    __syncthreads();
    push_trace(d, inst, S, pn);                                        // :184
    if (pn <= d.tol_direction) {                                       // :187-196
        if (fr) {
            if (threadIdx.x == 0) { S.fr = 0; S.iter += 1; S.stage = ST_TOP; S.budget -= 1; }
        } else {
            if (threadIdx.x == 0) S.ret = 0;
            __syncthreads();
            finalize(d, inst, S, x);
        }
        return;
    }
    if (S.prim_infeas <= d.tol_infeas && S.dual_infeas <= d.tol_residual && !isapprox_d(S.Delta, pn) &&
        !fr) {                                                       // :198-204
        if (threadIdx.x == 0) S.ret = 0;
        __syncthreads();
        finalize(d, inst, S, x);
        return;
    }
    // do_step! :515-579
    for (int j = threadIdx.x; j < d.n; j += TPB) tmpx[j] = x[j] + ps[j];
    __syncthreads();
    __shared__ double fsh;
    acopf_eval(d, inst, tmpx, 1.0, nullptr, &fsh, nullptr, tmpE, nullptr, nullptr);
    __syncthreads();
    const double c_k = viol1(d, tmpE, gL, gU, tmpx, xL, xU);
    const double phi_k = fr ? c_k : fsh + S.mu * c_k;
    double ared = S.phi - phi_k, pred = 1.0, q0 = 0.0;
    if (!fr) {
        q0 = S.mu * viol1(d, E, gL, gU, x, xL, xU);                    // compute_qmodel(sqp, false)
        const double qk = qmodel_step(d, inst, S, ps, x, df, E, jv, hv, gL, gU, xL, xU, tmpx, tmpE);
        pred = q0 - qk;
    }
    const double rho = ared / pred;
    if (ared > 0 && rho > 0) {                                         // :530-538
        accept_step(d, x, lam, mxL, mxU, ps, plam, pmxL, pmxU);
        if (threadIdx.x == 0) {
            if (isapprox_d(S.Delta, pn)) S.Delta = fmin(2 * S.Delta, 1e8);
            S.step_acceptance = 1;
        }
    } else {
        if (d.use_soc && c_k > 0 && !fr) {                           // :544-549 -> sub_optimize_soc! :341-360
            for (int j = threadIdx.x; j < d.n; j += TPB) tmpx[j] = x[j] + ps[j];
            __syncthreads();
            acopf_eval(d, inst, tmpx, 1.0, nullptr, nullptr, nullptr, Esoc, nullptr, nullptr);
            __syncthreads();
            double *bE = d.bE + om;
            for (int i = threadIdx.x; i < d.m; i += TPB) {
                double jp = 0.0;
                for (int k = d.jrowptr[i]; k < d.jrowptr[i + 1]; ++k) jp += jv[d.jrslot[k]] * ps[d.jrcol[k]];
                Esoc[i] -= jp;
                bE[i] = Esoc[i];
            }
            if (threadIdx.x == 0) {
                qp_request(I, SQPHIP_MODE_SOC, S.Delta, S.mu);
                S.q0 = q0; S.pnorm = pn; S.stage = ST_SOC;
            }
            return;
        }
        if (threadIdx.x == 0) {                                        // :574-577
            S.Delta = fmax(0.5 * fmin(S.Delta, pn), 0.1 * d.tol_direction);
            S.step_acceptance = 0;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (S.fr && S.step_acceptance) S.fr = 0;                       // :209-211
        S.iter += 1;                                                   // :213
        S.stage = ST_TOP; S.budget -= 1;
    }
}

// second half of do_step! for instances that requested a second-order correction (:551-572)
static __device__ __forceinline__ void b_sqp_soc_finish(const DV &d)
{
    const int inst = blockIdx.x;
    SQP_PTRS
    const bool go = !(S.done || S.stage != ST_SOC || !qp_final(d, inst));
    const int fr = S.fr;
    __syncthreads();
    if (!go) return;
    const double *op = d.op + on;
    for (int j = threadIdx.x; j < d.n; j += TPB) { psoc[j] = ps[j] + op[j]; tmpx[j] = x[j] + ps[j] + op[j]; }
    __syncthreads();
    __shared__ double fsh;
    acopf_eval(d, inst, tmpx, 1.0, nullptr, &fsh, nullptr, tmpE, nullptr, nullptr);
    __syncthreads();
    const double c_s = viol1(d, tmpE, gL, gU, tmpx, xL, xU);
    const double phi_soc = fr ? c_s : fsh + S.mu * c_s;
    const double ared = S.phi - phi_soc;
    const double qs = qmodel_step(d, inst, S, psoc, x, df, E, jv, hv, gL, gU, xL, xU, tmpx, tmpE);
    const double pred = S.q0 - qs;
    const double rho = ared / pred;
    if (threadIdx.x == 0) { S.n_qp++; S.it_ipm += I.ipm_iters; S.tot_ipm += I.ipm_iters; S.tot_fac += I.n_factor; S.tot_sol += I.n_solve; book_mode(S, I); }
    if (ared > 0 && rho > 0) {
        accept_step(d, x, lam, mxL, mxU, psoc, plam, pmxL, pmxU);
        if (threadIdx.x == 0) S.step_acceptance = 1;
    } else if (threadIdx.x == 0) {
        S.Delta = fmax(0.5 * fmin(S.Delta, S.pnorm), 0.1 * d.tol_direction);
        S.step_acceptance = 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (S.fr && S.step_acceptance) S.fr = 0;
        S.iter += 1;
        S.stage = ST_TOP; S.budget -= 1;
    }
}

__global__ void k_sqp_budget(DV d, int budget)
{
    for (int i = threadIdx.x; i < d.B; i += blockDim.x) d.sst[i].budget = budget;
}

__global__ void k_sqp_count(DV d, int *host_slot)
{
    // counters[2] = instances that still have work in this run, [3] = pending sub-problem starts; host_slot: the same
    // two words in pinned host memory (written from here: one launch per sweep less than a device-to-host copy behind it)
    // host_slot[1] (round 4): instances waiting for a refinement solve (PH_RESOLVE) -- with the monotone rule the second solve
    // slot of a sweep is launched only when the host has seen one (ipm_sweep, Ctx::want_resolve)
    int nb = 0, ns = 0, nr = 0;
    for (int i = threadIdx.x; i < d.B; i += blockDim.x) {
        // transitions on a side stream: the hand-over between the two sides (ctx.hpp, the PH_ enum); both streams are
        // quiet here -- this launch is behind the event of the side job, the next side job is behind this launch's
        if (d.side == 2) {
            const int ph = d.phase[i];
            if (ph == PH_DONE2) d.phase[i] = PH_DONE;
            else if (ph == PH_PEND) d.phase[i] = PH_FACTOR;
        }
        const SqpState &S = d.sst[i];
        if (!S.done && (S.budget > 0 || S.stage != ST_TOP)) ++nb;
        if (d.ist[i].start) ++ns;
        if (d.phase[i] == PH_RESOLVE) ++nr;
    }
    __shared__ int a[64], b[64], c[64];
    a[threadIdx.x] = nb; b[threadIdx.x] = ns; c[threadIdx.x] = nr;
    __syncthreads();
    if (threadIdx.x == 0) {
        int s0 = 0, s1 = 0, s2 = 0;
        for (int k = 0; k < (int)blockDim.x; ++k) { s0 += a[k]; s1 += b[k]; s2 += c[k]; }
        d.counters[2] = s0; d.counters[3] = s1;
        if (host_slot) { host_slot[0] = s0; host_slot[1] = s2; __threadfence_system(); }
    }
}

// scenario queue: every slot starts "terminated, nothing to file" and draws its first scenario in the first sweep
__global__ void k_sqp_stream_arm(DV d)
{
    for (int i = threadIdx.x; i < d.B; i += blockDim.x) {
        d.sst[i].done = 1; d.sst[i].stage = ST_DONE; d.stream.slot_scen[i] = -2;
    }
}

void sqp_stream_arm(Ctx &C)
{
    hipLaunchKernelGGL(k_sqp_reset, dim3(C.d.B), dim3(TPB), 0, C.stream, C.d);
    hipLaunchKernelGGL(k_sqp_stream_arm, dim3(1), dim3(256), 0, C.stream, C.d);
    SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
}

// slots that found the queue empty (-1) look again: ids may have been appended since (sqphip_sqp_stream_append)
__global__ void k_sqp_stream_rearm(DV d)
{
    for (int i = threadIdx.x; i < d.B; i += blockDim.x)
        if (d.stream.slot_scen[i] == -1) d.stream.slot_scen[i] = -2;
}

void sqp_stream_rearm(Ctx &C)
{
    hipLaunchKernelGGL(k_sqp_stream_rearm, dim3(1), dim3(256), 0, C.stream, C.d);
    SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
}

void sqp_reset(Ctx &C)
{
    hipLaunchKernelGGL(k_sqp_reset, dim3(C.d.B), dim3(TPB), 0, C.stream, C.d);
    SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
}

// Every instance performs up to `max_outer` more outer iterations of run! (0 = until it terminates).
// Continuous batching: one fixed kernel sequence per sweep; every kernel is gated on the per-instance
// stage / phase, so an instance whose sub-problem has converged goes through its merit step and into
// its next sub-problem while the others are still iterating -- no instance waits for the slowest.
static void sqp_run_lane(Ctx &C, int max_outer);

// With instance groups (Ctx::lanes) every group runs the same loop on its own stream from its own host thread;
// instances never interact, so the results do not depend on the grouping.
void sqp_run(Ctx &C, int max_outer)
{
    if (C.lanes.empty()) { sqp_run_lane(C, max_outer); return; }
    std::vector<std::thread> th;
    std::vector<std::string> errs(C.lanes.size());
    // nothing may leave a lane thread as an exception (std::terminate would take the host application down), and a
    // failure to start thread k must not destroy the joinable threads 0..k-1
    try {
        for (size_t g = 0; g < C.lanes.size(); ++g)
            th.emplace_back([&, g] {
                try {
                    SQPHIP_HIP_OK(hipSetDevice(C.opt.device));
                    sqp_run_lane(*C.lanes[g], max_outer);
                } catch (const std::string &e) { errs[g] = e; }
                catch (const std::bad_alloc &) { errs[g] = "sqphip: out of host memory in an instance group"; }
                catch (const std::exception &e) { errs[g] = std::string("sqphip: instance group: ") + e.what(); }
                catch (...) { errs[g] = "sqphip: unknown exception in an instance group"; }
            });
    } catch (...) {
        for (auto &t : th) t.join();
        throw std::string("sqphip: could not start the instance-group threads");
    }
    for (auto &t : th) t.join();
    for (auto &e : errs) if (!e.empty()) throw e;
}

static void sqp_run_lane(Ctx &C, int max_outer)
{
    DV &d = C.d;
    hipStream_t s = C.stream;
    const dim3 gB(d.B), bT(TPB);
    C.run_sweep = 0;                     // (the first sweep of a run always carries the transitions: ipm_sweep)
    C.want_resolve = true;               // (... and the refinement slot, until the first counter has come back)
    // transitions on a side stream (ctx.hpp, the PH_ enum; ipm_sweep): monotone rule, sparse path, one-workgroup vector stages
    C.side_on = C.side_mode && d.sparse && !d.flat && d.ipm_corrector == 0 && d.B >= 8;
    if (C.side_on && !C.side) {
        SQPHIP_HIP_OK(hipStreamCreateWithFlags(&C.side, hipStreamNonBlocking));
        for (auto &e : C.evS) SQPHIP_HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (auto &e : C.evC) SQPHIP_HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    d.side = C.side_on ? 2 : 0;
    struct SideOff { Ctx &C; ~SideOff() { C.d.side = 0; C.side_on = false; C.d.spec_mode = C.spec_mode0; } } side_off{C};     // (other entry points run in line)
    hipLaunchKernelGGL(k_sqp_budget, dim3(1), dim3(64), 0, s, d, max_outer > 0 ? max_outer : 0x3fffffff);
    hipLaunchKernelGGL(k_sqp_begin, gB, bT, 0, s, d);
    // The "anyone left?" counter of sweep k is read while sweep k + 1 is already queued: the stream never runs dry
    // behind a host round trip.  The price is one sweep of gated-off kernels after the last instance has finished.
    static const bool sweep_log = getenv("SQPHIP_SWEEP_LOG") != nullptr;    // instances with work left, per sweep
    static const bool lockstep = getenv("SQPHIP_SWEEP_LOCKSTEP") != nullptr; // experiment switch: read before queueing
    hipEvent_t ev[2];
    for (auto &e : ev) SQPHIP_HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    // experiment aid (SQPHIP_HOST_STATS): where the host thread of this group spends its time -- queueing a sweep, or waiting for
    // the counter of the sweep before last; a wait that returns at once means the stream may have run dry behind the host
    static const bool host_stats = getenv("SQPHIP_HOST_STATS") != nullptr;
    double hs_queue = 0.0, hs_wait = 0.0; long hs_sweeps = 0, hs_nowait = 0;
    auto left_after = [&](long k) {          // instances with work left after sweep k
        const auto w0 = std::chrono::steady_clock::now();
        SQPHIP_HIP_OK(hipEventSynchronize(ev[k & 1]));
        if (host_stats) {
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - w0).count();
            hs_wait += us; if (us < 5.0) ++hs_nowait;
        }
        SQPHIP_HIP_OK(hipGetLastError());   // a failed launch anywhere in the sweep surfaces here
        const int left = C.h_counters[2 + 2 * (k & 1)];
        C.want_resolve = C.h_counters[3 + 2 * (k & 1)] > 0;      // (refinement solves pending after sweep k: the next sweep queued carries the slot)
        if (C.spec_tail > 0) d.spec_mode = left <= C.spec_tail ? 1 : 0;     // second shift per sweep in the tail of the run (api.hip)
        if (sweep_log) fprintf(stderr, "%d%c", left, (k % 32) == 31 ? '\n' : ' ');
        return left;
    };
    try {
        for (long sweep = 0; sweep < 100000000L; ++sweep) {
            const auto q0 = std::chrono::steady_clock::now();
            ipm_sweep(C, /*sqp_level=*/true);
            int *slot = nullptr;                 // device view of the pinned words this sweep reports into
            SQPHIP_HIP_OK(hipHostGetDevicePointer((void **)&slot, C.h_counters + 2 + 2 * (sweep & 1), 0));
            if (C.side_on && sweep > 0) SQPHIP_HIP_OK(hipStreamWaitEvent(s, C.evS[sweep & 3], 0));     // the side job of this sweep
            hipLaunchKernelGGL(k_sqp_count, dim3(1), dim3(64), 0, s, d, slot);
            if (C.side_on) SQPHIP_HIP_OK(hipEventRecord(C.evC[sweep & 3], s));
            SQPHIP_HIP_OK(hipEventRecord(ev[sweep & 1], s));
            if (host_stats) { hs_queue += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - q0).count(); ++hs_sweeps; }
            if (C.tm.pending_trailing.size() > 4096) C.tm.flush();
            if (lockstep) { if (left_after(sweep) == 0) break; continue; }
            if (sweep >= 1 && left_after(sweep - 1) == 0) break;
        }
        SQPHIP_HIP_OK(hipStreamSynchronize(s));
        if (C.side_on) SQPHIP_HIP_OK(hipStreamSynchronize(C.side));
        if (host_stats && hs_sweeps > 0)
            fprintf(stderr, "sqphip: group of %d: %ld sweeps, host queues a sweep in %.1f us, waits %.1f us per sweep, %ld waits returned at once\n",
                    d.B, hs_sweeps, hs_queue / hs_sweeps, hs_wait / hs_sweeps, hs_nowait);
    } catch (...) {
        for (auto &e : ev) hipEventDestroy(e);
        throw;
    }
    for (auto &e : ev) hipEventDestroy(e);
    if (const char *e = getenv("SQPHIP_EMPTY_SWEEPS")) {      // experiment: wall time of a sweep with every kernel gated off
        const int n = atoi(e);
        C.side_on = false; d.side = 0;       // (the experiment's sweeps run in line)
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < n; ++k) ipm_sweep(C, true);
        SQPHIP_HIP_OK(hipStreamSynchronize(s));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        fprintf(stderr, "sqphip: %d gated-off sweeps of %d instances: %.1f us each\n", n, d.B, us / (n > 0 ? n : 1));
        C.n_sweeps -= n;
    }
}

// Scenario queue (ctx.hpp StreamDev): a slot whose run has terminated files its result under its scenario id, takes the
// next id, loads that scenario and runs the prologue of run! -- all inside the stage kernel of the sweep in which the
// run ended, so the slot never idles while scenarios are left.
static __device__ __forceinline__ void b_sqp_stream(const DV &d)
{
    const int inst = blockIdx.x;
    SQP_PTRS
    const StreamDev &Q = d.stream;
    const int cur = Q.slot_scen[inst];
    const bool go = S.done && cur != -1;
    __syncthreads();                 // gate read by every thread before any write
    if (!go) return;
    if (cur >= 0) {
        double *rx = Q.rx + (long)cur * d.n;
        for (int j = threadIdx.x; j < d.n; j += TPB) rx[j] = x[j];
        if (threadIdx.x == 0) { Q.robj[cur] = S.obj_val; Q.rstat[cur] = S.ret; Q.riter[cur] = S.iter; }
    }
    __shared__ int nxt;
    if (threadIdx.x == 0) {
        const int k = atomicAdd(Q.next, 1);
        nxt = k < *Q.qend ? Q.qids[k] : -1;
    }
    __syncthreads();
    const int sc = nxt;
    if (sc < 0) {
        if (threadIdx.x == 0) Q.slot_scen[inst] = -1;
        return;
    }
    {
        double *xLw = d.xL + on, *xUw = d.xU + on, *gLw = d.gL + om, *gUw = d.gU + om, *x0w = d.x0 + on;
        double *ohm = d.br_ohm + (long)inst * d.nl * 12, *c2 = d.c2 + (long)inst * d.ng, *c1 = d.c1 + (long)inst * d.ng;
        const double *sxL = Q.xL + (long)sc * d.n, *sxU = Q.xU + (long)sc * d.n, *sx0 = Q.x0 + (long)sc * d.n;
        const double *sgL = Q.gL + (long)sc * d.m, *sgU = Q.gU + (long)sc * d.m;
        const double *so = Q.ohm + (long)sc * d.nl * 12, *s2 = Q.c2 + (long)sc * d.ng, *s1 = Q.c1 + (long)sc * d.ng;
        for (int j = threadIdx.x; j < d.n; j += TPB) { xLw[j] = sxL[j]; xUw[j] = sxU[j]; x0w[j] = sx0[j]; }
        for (int i = threadIdx.x; i < d.m; i += TPB) { gLw[i] = sgL[i]; gUw[i] = sgU[i]; }
        for (int k = threadIdx.x; k < 12 * d.nl; k += TPB) ohm[k] = so[k];
        for (int g = threadIdx.x; g < d.ng; g += TPB) { c2[g] = s2[g]; c1[g] = s1[g]; }
    }
    __syncthreads();
    reset_instance(d, inst, true);
    if (threadIdx.x == 0) Q.slot_scen[inst] = sc;
    __syncthreads();
    b_sqp_begin(d);
}

// SQP-level stages of a sweep in dependency order, one kernel (one workgroup owns one instance: its stages run one
// after the other, a barrier in between publishes the stage word thread 0 wrote; every stage keeps its own gate)
__global__ __launch_bounds__(TPB, SQPHIP_VEC_WAVES_PER_EU) void k_sqp_stage(DV d)
{
    b_sqp_lp_finish(d);
    __syncthreads();
    b_sqp_mid(d);
    __syncthreads();
    if (d.use_soc) { b_sqp_soc_finish(d); __syncthreads(); }
    b_sqp_top(d);
    if (d.stream.M > 0) { __syncthreads(); b_sqp_stream(d); }
}

void sqp_stage_kernels(Ctx &C, hipStream_t s, const DV &d)          // called from ipm_sweep
{
    hipLaunchKernelGGL(k_sqp_stage, dim3(d.B), dim3(TPB), 0, s, d);
}

// ---------------------------------------------------------------------------------------------
// Merit / acceptance reductions for the drop-in path: operands are staged in instance 0's vectors
// (x, E, df, lambda, mult_x_U, mult_x_L, pstep, jcoo, hcoo) by the host wrapper.
//   op 0 norm_violations (common.jl:54-77)      op 1 KT_residuals (common.jl:14-23)
//   op 2 norm_complementarity (common.jl:30-47) op 3 compute_phi (sqp.jl:170-183)
//   op 4 compute_qmodel (sqp_trust_region.jl:487-508)  op 5 compute_derivative (merit.jl:15, sqp.jl:203-212)
__global__ __launch_bounds__(TPB) void k_merit(DV d, int op, double a0, double a1, int flag, double *out)
{
    const int inst = 0;
    SQP_PTRS
    double r = 0.0;
    if (op == 0 || op == 2) {
        double acc = 0.0, den = 0.0;
        if (op == 0) {
            for (int i = threadIdx.x; i < d.m; i += TPB) {
                double v = 0.0;
                if (E[i] > gU[i]) v = E[i] - gU[i]; else if (E[i] < gL[i]) v = gL[i] - E[i];
                acc = flag == 1 ? acc + v : (flag == 2 ? acc + v * v : fmax(acc, v));
            }
            for (int j = threadIdx.x; j < d.n; j += TPB) {
                double v = 0.0;
                if (x[j] > xU[j]) v = x[j] - xU[j]; else if (x[j] < xL[j]) v = xL[j] - x[j];
                acc = flag == 1 ? acc + v : (flag == 2 ? acc + v * v : fmax(acc, v));
            }
        } else {
            for (int i = threadIdx.x; i < d.m; i += TPB) {
                double c = 0.0;
                if (gL[i] != gU[i]) { c = fmin(E[i] - gL[i], gU[i] - E[i]) * lam[i]; den += lam[i] * lam[i]; }
                c = fabs(c);
                acc = flag == 1 ? acc + c : (flag == 2 ? acc + c * c : fmax(acc, c));
            }
        }
        acc = flag == 0 ? block_reduce<OpMax>(acc) : block_reduce<OpSum>(acc);
        if (flag == 2) acc = sqrt(acc);
        if (op == 2) { den = block_reduce<OpSum>(den); acc = acc / (1.0 + sqrt(den)); }
        r = acc;
    } else if (op == 1) {
        gather_csc(d, jcoo, hcoo, jv, nullptr);
        __syncthreads();
        r = kt_residuals(d, df, lam, mxU, mxL, jv, 1.0, tmpE);
    } else if (op == 3) {
        const double v = viol1(d, E, gL, gU, x, xL, xU);
        r = flag ? v : a0 + a1 * v;          // a0 = f(x + alpha p), a1 = mu, flag = feasibility restoration
    } else if (op == 4) {
        SqpState tmp = S;
        tmp.mu = a1;
        if (flag) {
            gather_csc(d, jcoo, hcoo, jv, d.nnzh_coo ? hv : nullptr);
            __syncthreads();
            r = qmodel_step(d, inst, tmp, ps, x, df, E, jv, hv, gL, gU, xL, xU, tmpx, tmpE);
        } else {
            r = a1 * viol1(d, E, gL, gU, x, xL, xU);
        }
    } else if (op == 5) {
        double dfp = 0.0, cv = 0.0;
        for (int j = threadIdx.x; j < d.n; j += TPB) dfp += df[j] * ps[j];
        for (int i = threadIdx.x; i < d.m; i += TPB) cv += fmax(0.0, fmax(E[i] - gU[i], gL[i] - E[i]));
        dfp = block_reduce<OpSum>(dfp); cv = block_reduce<OpSum>(cv);
        r = dfp - a1 * cv;
    } else if (op == 6) {
        // compute_derivative(sqp), sqp.jl:190-213 over merit.jl:13-17.  flag bit 0: feasibility restoration (dfp =
        // sum of the slacks staged in oslack), bit 1: vector penalty staged in plam
        const bool fr = flag & 1, vec = flag & 2;
        const double *slack = d.oslack;
        double dfp = 0.0, cv = 0.0;
        if (fr) { for (int k = threadIdx.x; k < 2 * d.m; k += TPB) dfp += slack[k]; }
        else for (int j = threadIdx.x; j < d.n; j += TPB) dfp += df[j] * ps[j];
        for (int i = threadIdx.x; i < d.m; i += TPB) {
            double v = fmax(0.0, fmax(E[i] - gU[i], gL[i] - E[i]));
            if (fr) { const double lhs = E[i] - v; v = fmax(0.0, fmax(lhs - gU[i], gL[i] - lhs)); }
            cv += vec ? plam[i] * v : v;
        }
        dfp = block_reduce<OpSum>(dfp); cv = block_reduce<OpSum>(cv);
        r = vec ? dfp - cv : dfp - a1 * cv;
    } else if (op == 7) {
        // compute_mu_rule1! / 2! / 3! (sqp_line_search.jl:270-294): a0 = rho, flag = rule | (iter == 1) << 4;
        // mu[m] staged in plam (updated in place), lambda in lam
        const int rule = flag & 15, first = flag >> 4;
        double dfp = 0.0, php = 0.0;
        gather_csc(d, jcoo, hcoo, jv, d.nnzh_coo ? hv : nullptr);
        __syncthreads();
        for (int j = threadIdx.x; j < d.n; j += TPB) {
            double hp = 0.0;
            for (int k = d.hcolptr[j]; k < d.hcolptr[j + 1]; ++k) hp += hv[k] * ps[d.hrowval[k]];
            dfp += df[j] * ps[j]; php += 0.5 * ps[j] * hp;
        }
        dfp = block_reduce<OpSum>(dfp); php = block_reduce<OpSum>(php);
        const double v1 = viol1(d, E, gL, gU, x, xL, xU);
        const double t = (dfp + fmax(php, 0.0)) / fmax((1.0 - a0) * v1, 1.0e-8);
        for (int i = threadIdx.x; i < d.m; i += TPB) {
            double mu = plam[i];
            if (rule == 1) { mu = fmax(mu, t); mu = fmax(mu, fabs(lam[i])); }
            else if (rule == 2) mu = first ? t : fmax(mu, fabs(lam[i]));
            else mu = fmax(mu, fabs(lam[i]));
            plam[i] = mu;
        }
        r = t;
    }
    if (threadIdx.x == 0) *out = r;
}

// compute_alpha (sqp_line_search.jl:303-334) on the device: x and p of instance `inst` are staged in d.x / d.pstep;
// out[0] = alpha, out[1] = is_valid, out[2] = merit evaluations.  One workgroup; the merit function is
// compute_phi (sqp.jl:170-183) over the device callbacks, its norms reduced by wave butterflies + an LDS exchange.
__global__ __launch_bounds__(TPB) void k_armijo(DV d, int inst, double mu, double phi0, double D, double eta, double tau,
                                                double min_alpha, int fr, double *out)
{
    SQP_PTRS
    const double pn = norm_inf(ps, d.n);
    double alpha = 1.0;
    int valid = 1, nev = 0;
    if (!(pn <= d.tol_direction)) {
        __shared__ double fsh;
        for (;;) {
            for (int j = threadIdx.x; j < d.n; j += TPB) tmpx[j] = x[j] + alpha * ps[j];
            __syncthreads();
            acopf_eval(d, inst, tmpx, 1.0, nullptr, &fsh, nullptr, tmpE, nullptr, nullptr);
            __syncthreads();
            const double v = viol1(d, tmpE, gL, gU, tmpx, xL, xU);
            const double phi = fr ? v : fsh + mu * v;
            ++nev;
            if (!(phi > phi0 + eta * alpha * D)) break;
            if (alpha < min_alpha) { valid = 0; break; }     // the step size can become too small
            alpha *= tau;
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) { out[0] = alpha; out[1] = valid; out[2] = nev; }
}

void armijo_eval(Ctx &C, int inst, double mu, double phi0, double D, double eta, double tau, double min_alpha, int fr,
                 double *out3_host)
{
    double *o = C.d.wN + (size_t)inst * C.d.Npad;     // scratch slots
    hipLaunchKernelGGL(k_armijo, dim3(1), dim3(TPB), 0, C.stream, C.d, inst, mu, phi0, D, eta, tau, min_alpha, fr, o);
    SQPHIP_HIP_OK(hipMemcpyAsync(out3_host, o, 3 * sizeof(double), hipMemcpyDeviceToHost, C.stream));
    SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
}

void merit_eval(Ctx &C, int op, double a0, double a1, int flag, double *out_host)
{
    double *o = C.d.wN;     // scratch scalar slot
    hipLaunchKernelGGL(k_merit, dim3(1), dim3(TPB), 0, C.stream, C.d, op, a0, a1, flag, o);
    SQPHIP_HIP_OK(hipMemcpyAsync(out_host, o, sizeof(double), hipMemcpyDeviceToHost, C.stream));
    SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
}

}  // namespace sqphip
