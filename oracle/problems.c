/* problems.c -- NLP instances for the oracle.  TEST INFRASTRUCTURE ONLY.
 * toy:     /root/reference/test/ext_solver.jl:14-28 laid out as MOI_wrapper.jl:1081-1199 would
 * readme1: /root/reference/README.md:18-21,37-39
 * hs071:   Hock-Schittkowski 71 as in MathOptInterface's nonlinear tests (not vendored)
 * acopf:   PowerModels ACP build_opf shape (test/opf.jl:5-9), SURVEY.md Appendix B, over the
 *          arrays produced by sqpsolver.jl_amd/acopf_synth.py */
#include "sqp_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

struct ora_problem {
    ora_nlp nlp;
    double *x0;
    int64_t *jrow, *jcol, *hrow, *hcol;
    double *xL, *xU, *gL, *gU;
    /* acopf */
    int nb, ng, nl, ref_bus;
    int32_t *f_bus, *t_bus, *gen_bus, *bal_ptr, *bal_colP, *bal_colQ;
    int ndc; double *dc_loss1;    /* HVDC lines: variables p_f, p_t, q_f, q_t behind all others, one loss row each at the end */
    int nsh; int32_t *sh_bus; double *sh_gs, *sh_bs;   /* bus shunts: + gs vm^2 (P row), - bs vm^2 (Q row) */
    double *ohm, *c2, *c1, *bal_coef;      /* ohm[nl][12]: (A, Bc, Bs) of p_f, q_f, p_t, q_t per branch */
    /* W-space form (acwr): bus pairs i < j, pair and orientation of every branch, tan of the pairs' angle limits, shunts per bus */
    int nbp; int32_t *bp_i, *bp_j, *br_bp; double *br_sig, *bp_tmin, *bp_tmax, *gsb, *bsb;
    /* synthetic dense-Hessian NLP (ora_problem_dense): Q [n][n] symmetric, A [m][n] row-major, linear cost, quartic weight */
    double *dnQ, *dnA, *dnc, dn_kappa;
};

static int64_t *i64dup(const int64_t *s, int64_t k)
{ int64_t *d = (int64_t *)malloc(sizeof(int64_t) * (size_t)(k + 1)); if (k) memcpy(d, s, sizeof(int64_t) * (size_t)k); return d; }
static int32_t *i32dup(const int32_t *s, int64_t k)
{ int32_t *d = (int32_t *)malloc(sizeof(int32_t) * (size_t)(k + 1)); if (k) memcpy(d, s, sizeof(int32_t) * (size_t)k); return d; }
static double *ddup(const double *s, int64_t k)
{ double *d = (double *)malloc(sizeof(double) * (size_t)(k + 1)); if (k) memcpy(d, s, sizeof(double) * (size_t)k); return d; }

static ora_problem *mk(int64_t n, int64_t m, int64_t nlin, int64_t nnzj, const int64_t *jr,
                       const int64_t *jc, int64_t nnzh, const int64_t *hr, const int64_t *hc,
                       const double *xL, const double *xU, const double *gL, const double *gU,
                       const double *x0)
{
    ora_problem *P = (ora_problem *)calloc(1, sizeof(ora_problem));
    P->jrow = i64dup(jr, nnzj); P->jcol = i64dup(jc, nnzj);
    P->hrow = i64dup(hr, nnzh); P->hcol = i64dup(hc, nnzh);
    P->xL = ddup(xL, n); P->xU = ddup(xU, n); P->gL = ddup(gL, m); P->gU = ddup(gU, m);
    P->x0 = ddup(x0, n);
    P->nlp.n = n; P->nlp.m = m; P->nlp.num_linear = nlin; P->nlp.nnzj = nnzj; P->nlp.nnzh = nnzh;
    P->nlp.jrow = P->jrow; P->nlp.jcol = P->jcol; P->nlp.hrow = P->hrow; P->nlp.hcol = P->hcol;
    P->nlp.xL = P->xL; P->nlp.xU = P->xU; P->nlp.gL = P->gL; P->nlp.gU = P->gU;
    P->nlp.ud = P;
    return P;
}
const ora_nlp *ora_problem_nlp(const ora_problem *p) { return &p->nlp; }
/* the problem without second derivatives: what the reference sees when the evaluator offers no :Hess feature
 * (MOI_wrapper.jl:1092-1103,1178: eval_h === nothing, hessian_type "none"; sqp.jl:92 then never fills the Hessian and
 * subproblem_JuMP.jl:137-140 gives every sub-problem a linear objective: sequential linear programming) */
void ora_problem_drop_hessian(ora_problem *p) { p->nlp.nnzh = 0; p->nlp.eval_h = NULL; }
const double *ora_problem_x0(const ora_problem *p) { return p->x0; }
void ora_problem_destroy(ora_problem *P)
{
    if (!P) return;
    void *ptrs[] = { P->x0, P->jrow, P->jcol, P->hrow, P->hcol, P->xL, P->xU, P->gL, P->gU,
        P->f_bus, P->t_bus, P->gen_bus, P->bal_ptr, P->bal_colP, P->bal_colQ, P->ohm,
        P->c2, P->c1, P->bal_coef, P->sh_bus, P->sh_gs, P->sh_bs, P->dc_loss1,
        P->bp_i, P->bp_j, P->br_bp, P->br_sig, P->bp_tmin, P->bp_tmax, P->gsb, P->bsb, P->dnQ, P->dnA, P->dnc };
    for (size_t i = 0; i < sizeof(ptrs) / sizeof(ptrs[0]); ++i) free(ptrs[i]);
    free(P);
}

/* ------------------------------------------------------------------ toy (ext_solver.jl) */
static double toy_f(void *u, const double *x) { (void)u; return x[0] * x[0] + x[0]; }
static void toy_df(void *u, const double *x, double *g) { (void)u; g[0] = 2 * x[0] + 1; g[1] = 0; }
static void toy_g(void *u, const double *x, double *g)
{ (void)u; g[0] = x[0]; g[1] = x[0] * x[0] - x[0] - 2; g[2] = x[0] * x[1] - 1; g[3] = x[0] * x[1]; }
static void toy_jac(void *u, const double *x, double *v)
{ (void)u; v[0] = 1; v[1] = 2 * x[0] - 1; v[2] = x[1]; v[3] = x[0]; v[4] = x[1]; v[5] = x[0]; }
static void toy_h(void *u, const double *x, double s, const double *l, double *v)
{ (void)u; (void)x; v[0] = 2 * s; v[1] = 2 * l[1]; v[2] = l[2]; v[3] = l[3]; }

ora_problem *ora_problem_toy(void)
{
    int64_t jr[] = {1, 2, 3, 3, 4, 4}, jc[] = {1, 1, 1, 2, 1, 2};
    int64_t hr[] = {1, 1, 2, 2}, hc[] = {1, 1, 1, 1};
    double xL[] = {-INFINITY, -INFINITY}, xU[] = {INFINITY, INFINITY};
    double gL[] = {-2, 0, 0, 0}, gU[] = {INFINITY, 0, 0, INFINITY};
    double x0[] = {0, 0};
    ora_problem *P = mk(2, 4, 1, 6, jr, jc, 4, hr, hc, xL, xU, gL, gU, x0);
    P->nlp.eval_f = toy_f; P->nlp.eval_grad_f = toy_df; P->nlp.eval_g = toy_g;
    P->nlp.eval_jac_g = toy_jac; P->nlp.eval_h = toy_h;
    return P;
}

/* ------------------------------------------------------------------ README one-variable */
static double r1_f(void *u, const double *x) { (void)u; return x[0] * x[0] + x[0]; }
static void r1_df(void *u, const double *x, double *g) { (void)u; g[0] = 2 * x[0] + 1; }
static void r1_g(void *u, const double *x, double *g) { (void)u; g[0] = x[0] * x[0] - x[0] - 2; }
static void r1_jac(void *u, const double *x, double *v) { (void)u; v[0] = 2 * x[0] - 1; }
static void r1_h(void *u, const double *x, double s, const double *l, double *v)
{ (void)u; (void)x; v[0] = 2 * s; v[1] = 2 * l[0]; }

ora_problem *ora_problem_readme1(void)
{
    int64_t jr[] = {1}, jc[] = {1}, hr[] = {1, 1}, hc[] = {1, 1};
    double xL[] = {-INFINITY}, xU[] = {INFINITY}, gL[] = {0}, gU[] = {0}, x0[] = {0};
    ora_problem *P = mk(1, 1, 0, 1, jr, jc, 2, hr, hc, xL, xU, gL, gU, x0);
    P->nlp.eval_f = r1_f; P->nlp.eval_grad_f = r1_df; P->nlp.eval_g = r1_g;
    P->nlp.eval_jac_g = r1_jac; P->nlp.eval_h = r1_h;
    return P;
}

/* ------------------------------------------------------------------ HS071 */
static double hs_f(void *u, const double *x) { (void)u; return x[0] * x[3] * (x[0] + x[1] + x[2]) + x[2]; }
static void hs_df(void *u, const double *x, double *g)
{
    (void)u;
    g[0] = x[3] * (2 * x[0] + x[1] + x[2]); g[1] = x[0] * x[3];
    g[2] = x[0] * x[3] + 1; g[3] = x[0] * (x[0] + x[1] + x[2]);
}
static void hs_g(void *u, const double *x, double *g)
{ (void)u; g[0] = x[0] * x[1] * x[2] * x[3]; g[1] = x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]; }
static void hs_jac(void *u, const double *x, double *v)
{
    (void)u;
    v[0] = x[1] * x[2] * x[3]; v[1] = x[0] * x[2] * x[3]; v[2] = x[0] * x[1] * x[3]; v[3] = x[0] * x[1] * x[2];
    for (int i = 0; i < 4; ++i) v[4 + i] = 2 * x[i];
}
static void hs_h(void *u, const double *x, double s, const double *l, double *v)
{
    (void)u;
    /* lower triangle row-major: (1,1),(2,1),(2,2),(3,1),(3,2),(3,3),(4,1),(4,2),(4,3),(4,4) */
    v[0] = s * 2 * x[3] + l[1] * 2;
    v[1] = s * x[3] + l[0] * x[2] * x[3];
    v[2] = l[1] * 2;
    v[3] = s * x[3] + l[0] * x[1] * x[3];
    v[4] = l[0] * x[0] * x[3];
    v[5] = l[1] * 2;
    v[6] = s * (2 * x[0] + x[1] + x[2]) + l[0] * x[1] * x[2];
    v[7] = s * x[0] + l[0] * x[0] * x[2];
    v[8] = s * x[0] + l[0] * x[0] * x[1];
    v[9] = l[1] * 2;
}
ora_problem *ora_problem_hs071(void)
{
    int64_t jr[] = {1, 1, 1, 1, 2, 2, 2, 2}, jc[] = {1, 2, 3, 4, 1, 2, 3, 4};
    int64_t hr[] = {1, 2, 2, 3, 3, 3, 4, 4, 4, 4}, hc[] = {1, 1, 2, 1, 2, 3, 1, 2, 3, 4};
    double xL[] = {1, 1, 1, 1}, xU[] = {5, 5, 5, 5};
    double gL[] = {25, 40}, gU[] = {INFINITY, 40}, x0[] = {1, 5, 5, 1};
    ora_problem *P = mk(4, 2, 0, 8, jr, jc, 10, hr, hc, xL, xU, gL, gU, x0);
    P->nlp.eval_f = hs_f; P->nlp.eval_grad_f = hs_df; P->nlp.eval_g = hs_g;
    P->nlp.eval_jac_g = hs_jac; P->nlp.eval_h = hs_h;
    return P;
}

/* ------------------------------------------------------------------ ACOPF (polar, ACP shape) */
/* Row k of branch l:  flow_k - F_k,  F_k = A_k v_self^2 + vf vt (Bc_k cos th + Bs_k sin th) */
static void ohm_coef(const ora_problem *P, int l, int k, double *A, double *Bc, double *Bs, int *self_t)
{
    /* twelve numbers per branch from the host (pi model, tap ratio and phase shift at the from end folded in:
     * sqpsolver.jl_amd/acopf_synth.py, Network.branch_coeffs) */
    const double *oc = P->ohm + 12 * (size_t)l + 3 * k;
    *A = oc[0]; *Bc = oc[1]; *Bs = oc[2]; *self_t = k >= 2;
}
#define IDX(P) \
    const int nb = (P)->nb, ng = (P)->ng, nl = (P)->nl; \
    const int VA = 0, VM = nb, PG = 2 * nb, QG = 2 * nb + ng, PF = 2 * nb + 2 * ng; \
    const int PT = PF + nl, QF = PF + 2 * nl, QT = PF + 3 * nl; \
    const int T0 = 2 * nl + 1 + 2 * nb, O0 = T0 + 2 * nl; \
    (void)VA; (void)VM; (void)PG; (void)QG; (void)PT; (void)QF; (void)QT; (void)T0; (void)O0;

static double ac_f(void *u, const double *x)
{
    const ora_problem *P = (const ora_problem *)u; IDX(P)
    double f = 0.0;
    for (int g = 0; g < ng; ++g) f += P->c2[g] * x[PG + g] * x[PG + g] + P->c1[g] * x[PG + g];
    return f;
}
static void ac_df(void *u, const double *x, double *gr)
{
    const ora_problem *P = (const ora_problem *)u; IDX(P)
    memset(gr, 0, sizeof(double) * (size_t)P->nlp.n);
    for (int g = 0; g < ng; ++g) gr[PG + g] = 2 * P->c2[g] * x[PG + g] + P->c1[g];
}
static void ac_g(void *u, const double *x, double *gv)
{
    const ora_problem *P = (const ora_problem *)u; IDX(P)
    const int own[4] = { PF, QF, PT, QT };
    for (int l = 0; l < nl; ++l) {
        double th = x[VA + P->f_bus[l]] - x[VA + P->t_bus[l]];
        gv[l] = th; gv[nl + l] = th;
    }
    gv[2 * nl] = x[VA + P->ref_bus];
    for (int i = 0; i < nb; ++i) {
        double sp = 0.0, sq = 0.0;
        for (int k = P->bal_ptr[i]; k < P->bal_ptr[i + 1]; ++k) {
            sp += P->bal_coef[k] * x[P->bal_colP[k]];
            sq += P->bal_coef[k] * x[P->bal_colQ[k]];
        }
        gv[2 * nl + 1 + 2 * i] = sp; gv[2 * nl + 2 + 2 * i] = sq;
    }
    for (int s = 0; s < P->nsh; ++s) {
        int i = P->sh_bus[s]; double vm = x[VM + i];
        gv[2 * nl + 1 + 2 * i] += P->sh_gs[s] * vm * vm; gv[2 * nl + 2 + 2 * i] -= P->sh_bs[s] * vm * vm;
    }
    for (int l = 0; l < nl; ++l) {
        gv[T0 + 2 * l] = x[PF + l] * x[PF + l] + x[QF + l] * x[QF + l];
        gv[T0 + 2 * l + 1] = x[PT + l] * x[PT + l] + x[QT + l] * x[QT + l];
        double vf = x[VM + P->f_bus[l]], vt = x[VM + P->t_bus[l]];
        double th = x[VA + P->f_bus[l]] - x[VA + P->t_bus[l]], C = cos(th), S = sin(th);
        for (int k = 0; k < 4; ++k) {
            double A, Bc, Bs; int st; ohm_coef(P, l, k, &A, &Bc, &Bs, &st);
            double vs = st ? vt : vf;
            gv[O0 + 4 * l + k] = x[own[k] + l] - (A * vs * vs + vf * vt * (Bc * C + Bs * S));
        }
    }
    for (int d = 0; d < P->ndc; ++d)           /* (1 - loss1) p_dc_f + p_dc_t  (= loss0 by the row bounds) */
        gv[O0 + 4 * nl + d] = (1.0 - P->dc_loss1[d]) * x[PF + 4 * nl + d] + x[PF + 4 * nl + P->ndc + d];
}
static void ac_jac(void *u, const double *x, double *v)
{
    const ora_problem *P = (const ora_problem *)u; IDX(P)
    int64_t o = 0;
    for (int rep = 0; rep < 2; ++rep) for (int l = 0; l < nl; ++l) { v[o++] = 1.0; v[o++] = -1.0; }
    v[o++] = 1.0;
    for (int i = 0; i < nb; ++i) {
        int s = P->bal_ptr[i], e = P->bal_ptr[i + 1];
        for (int k = s; k < e; ++k) v[o++] = P->bal_coef[k];
        for (int k = s; k < e; ++k) v[o++] = P->bal_coef[k];
    }
    for (int l = 0; l < nl; ++l) { v[o++] = 2 * x[PF + l]; v[o++] = 2 * x[QF + l]; }
    for (int l = 0; l < nl; ++l) { v[o++] = 2 * x[PT + l]; v[o++] = 2 * x[QT + l]; }
    for (int k = 0; k < 4; ++k)
        for (int l = 0; l < nl; ++l) {
            double vf = x[VM + P->f_bus[l]], vt = x[VM + P->t_bus[l]];
            double th = x[VA + P->f_bus[l]] - x[VA + P->t_bus[l]], C = cos(th), S = sin(th);
            double A, Bc, Bs; int st; ohm_coef(P, l, k, &A, &Bc, &Bs, &st);
            double T0v = Bc * C + Bs * S, T1 = -Bc * S + Bs * C, uu = vf * vt;
            v[o++] = 1.0;
            v[o++] = -(uu * T1);
            v[o++] = uu * T1;
            v[o++] = -((st ? 0.0 : 2 * A * vf) + vt * T0v);
            v[o++] = -((st ? 2 * A * vt : 0.0) + vf * T0v);
        }
    for (int s = 0; s < P->nsh; ++s) {
        double vm = x[VM + P->sh_bus[s]];
        v[o++] = 2 * P->sh_gs[s] * vm; v[o++] = -2 * P->sh_bs[s] * vm;
    }
    for (int d = 0; d < P->ndc; ++d) { v[o++] = 1.0 - P->dc_loss1[d]; v[o++] = 1.0; }
}
static void ac_h(void *u, const double *x, double sig, const double *lam, double *v)
{
    const ora_problem *P = (const ora_problem *)u; IDX(P)
    int64_t o = 0;
    for (int g = 0; g < ng; ++g) v[o++] = sig * 2 * P->c2[g];
    for (int l = 0; l < nl; ++l) { double w = 2 * lam[T0 + 2 * l]; v[o++] = w; v[o++] = w; }
    for (int l = 0; l < nl; ++l) { double w = 2 * lam[T0 + 2 * l + 1]; v[o++] = w; v[o++] = w; }
    for (int k = 0; k < 4; ++k) {
        double *blk = v + o + (int64_t)k * 10 * nl;
        for (int l = 0; l < nl; ++l) {
            double vf = x[VM + P->f_bus[l]], vt = x[VM + P->t_bus[l]];
            double th = x[VA + P->f_bus[l]] - x[VA + P->t_bus[l]], C = cos(th), S = sin(th);
            double A, Bc, Bs; int st; ohm_coef(P, l, k, &A, &Bc, &Bs, &st);
            double T0v = Bc * C + Bs * S, T1 = -Bc * S + Bs * C, uu = vf * vt;
            double w = -lam[O0 + 4 * l + k];       /* row = flow - F  =>  Hess(row) = -Hess(F) */
            blk[0 * nl + l] = w * (-uu * T0v);     /* va_f va_f */
            blk[1 * nl + l] = w * (uu * T0v);      /* va_t va_f */
            blk[2 * nl + l] = w * (-uu * T0v);     /* va_t va_t */
            blk[3 * nl + l] = w * (vt * T1);       /* vm_f va_f */
            blk[4 * nl + l] = w * (-vt * T1);      /* vm_f va_t */
            blk[5 * nl + l] = w * (st ? 0.0 : 2 * A); /* vm_f vm_f */
            blk[6 * nl + l] = w * (vf * T1);       /* vm_t va_f */
            blk[7 * nl + l] = w * (-vf * T1);      /* vm_t va_t */
            blk[8 * nl + l] = w * T0v;             /* vm_t vm_f */
            blk[9 * nl + l] = w * (st ? 2 * A : 0.0); /* vm_t vm_t */
        }
    }
    o += (int64_t)40 * nl;
    for (int s = 0; s < P->nsh; ++s) {
        int i = P->sh_bus[s];
        v[o++] = lam[2 * nl + 1 + 2 * i] * 2 * P->sh_gs[s] - lam[2 * nl + 2 + 2 * i] * 2 * P->sh_bs[s];
    }
}

ora_problem *ora_problem_acopf(int nb, int ng, int nl, const int32_t *f_bus,
                               const int32_t *t_bus, const double *ohm,
                               const int32_t *gen_bus, const double *c2,
                               const double *c1, const int32_t *bal_ptr, const int32_t *bal_colP,
                               const int32_t *bal_colQ, const double *bal_coef,
                               int64_t nnzj, const int64_t *jrow, const int64_t *jcol,
                               int64_t nnzh, const int64_t *hrow, const int64_t *hcol,
                               const double *xL, const double *xU, const double *gL,
                               const double *gU, int nsh, const int32_t *sh_bus, const double *sh_gs,
                               const double *sh_bs, int ndc, const double *dc_loss1)
{
    int64_t n = 2 * nb + 2 * ng + 4 * nl + 4 * ndc, m = 1 + 2 * nb + 8 * nl + ndc;
    double *x0 = (double *)calloc((size_t)n, sizeof(double));
    /* with bus shunts the balance rows carry a vm^2 term: only the angle and reference rows stay linear */
    ora_problem *P = mk(n, m, nsh > 0 ? 2 * nl + 1 : 2 * nl + 1 + 2 * nb, nnzj, jrow, jcol, nnzh, hrow, hcol, xL, xU, gL, gU, x0);
    free(x0);
    P->nb = nb; P->ng = ng; P->nl = nl;
    P->ref_bus = (int)(jcol[4 * nl] - 1);
    P->f_bus = i32dup(f_bus, nl); P->t_bus = i32dup(t_bus, nl); P->gen_bus = i32dup(gen_bus, ng);
    P->bal_ptr = i32dup(bal_ptr, nb + 1);
    int64_t nbal = bal_ptr[nb];
    P->bal_colP = i32dup(bal_colP, nbal); P->bal_colQ = i32dup(bal_colQ, nbal);
    P->bal_coef = ddup(bal_coef, nbal);
    P->ohm = ddup(ohm, 12 * (int64_t)nl);
    P->ndc = ndc; P->dc_loss1 = ddup(dc_loss1, ndc);
    P->nsh = nsh; P->sh_bus = i32dup(sh_bus, nsh); P->sh_gs = ddup(sh_gs, nsh); P->sh_bs = ddup(sh_bs, nsh);
    P->c2 = ddup(c2, ng); P->c1 = ddup(c1, ng);
    P->nlp.eval_f = ac_f; P->nlp.eval_grad_f = ac_df; P->nlp.eval_g = ac_g;
    P->nlp.eval_jac_g = ac_jac; P->nlp.eval_h = ac_h;
    return P;
}

/* ------------------------------------------------------------------ ACOPF (rectangular, ACR shape) */
/* PowerModels ACRPowerModel under the build_opf of /root/reference/examples/acopf/opf.jl:12-43 (what run_sqp_opf
 * instantiates, :46,:51), laid out by sqpsolver.jl_amd/acopf_synth.py, acr_layout: x = (vi, vr, pg, qg, flows, dc),
 * rows = vi[ref]; balance; vmin^2 <= vr^2+vi^2 and vr^2+vi^2 <= vmax^2 per bus; thermal; Ohm; dc losses.
 * F_k = A (vr_s^2 + vi_s^2) + Bc (vr_f vr_t + vi_f vi_t) + Bs (vi_f vr_t - vr_f vi_t): every row is quadratic. */
#define RIDX(P) \
    const int nb = (P)->nb, ng = (P)->ng, nl = (P)->nl; \
    const int VI = 0, VR = nb, PG = 2 * nb, PF = 2 * nb + 2 * ng; \
    const int PT = PF + nl, QF = PF + 2 * nl, QT = PF + 3 * nl; \
    const int V0 = 1 + 2 * nb, T0 = V0 + 2 * nb, O0 = T0 + 2 * nl; \
    (void)VI; (void)VR; (void)PG; (void)PT; (void)QF; (void)QT; (void)V0; (void)T0; (void)O0; (void)ng;

static void acr_g(void *u, const double *x, double *gv)
{
    const ora_problem *P = (const ora_problem *)u; RIDX(P)
    const int own[4] = { PF, QF, PT, QT };
    gv[0] = x[VI + P->ref_bus];
    for (int i = 0; i < nb; ++i) {
        double sp = 0.0, sq = 0.0;
        for (int k = P->bal_ptr[i]; k < P->bal_ptr[i + 1]; ++k) {
            sp += P->bal_coef[k] * x[P->bal_colP[k]];
            sq += P->bal_coef[k] * x[P->bal_colQ[k]];
        }
        gv[1 + 2 * i] = sp; gv[2 + 2 * i] = sq;
        double w = x[VR + i] * x[VR + i] + x[VI + i] * x[VI + i];
        gv[V0 + 2 * i] = w; gv[V0 + 2 * i + 1] = w;
    }
    for (int s = 0; s < P->nsh; ++s) {
        int i = P->sh_bus[s]; double w = x[VR + i] * x[VR + i] + x[VI + i] * x[VI + i];
        gv[1 + 2 * i] += P->sh_gs[s] * w; gv[2 + 2 * i] -= P->sh_bs[s] * w;
    }
    for (int l = 0; l < nl; ++l) {
        gv[T0 + 2 * l] = x[PF + l] * x[PF + l] + x[QF + l] * x[QF + l];
        gv[T0 + 2 * l + 1] = x[PT + l] * x[PT + l] + x[QT + l] * x[QT + l];
        const int fb = P->f_bus[l], tb = P->t_bus[l];
        double vrf = x[VR + fb], vif = x[VI + fb], vrt = x[VR + tb], vit = x[VI + tb];
        double cc = vrf * vrt + vif * vit, ss = vif * vrt - vrf * vit;
        for (int k = 0; k < 4; ++k) {
            double A, Bc, Bs; int st; ohm_coef(P, l, k, &A, &Bc, &Bs, &st);
            double w = st ? vrt * vrt + vit * vit : vrf * vrf + vif * vif;
            gv[O0 + 4 * l + k] = x[own[k] + l] - (A * w + Bc * cc + Bs * ss);
        }
    }
    for (int d = 0; d < P->ndc; ++d)
        gv[O0 + 4 * nl + d] = (1.0 - P->dc_loss1[d]) * x[PF + 4 * nl + d] + x[PF + 4 * nl + P->ndc + d];
}
static void acr_jac(void *u, const double *x, double *v)
{
    const ora_problem *P = (const ora_problem *)u; RIDX(P)
    int64_t o = 0;
    v[o++] = 1.0;
    for (int i = 0; i < nb; ++i) {
        int s = P->bal_ptr[i], e = P->bal_ptr[i + 1];
        for (int k = s; k < e; ++k) v[o++] = P->bal_coef[k];
        for (int k = s; k < e; ++k) v[o++] = P->bal_coef[k];
    }
    for (int i = 0; i < nb; ++i) { v[o++] = 2 * x[VR + i]; v[o++] = 2 * x[VI + i]; }
    for (int i = 0; i < nb; ++i) { v[o++] = 2 * x[VR + i]; v[o++] = 2 * x[VI + i]; }
    for (int l = 0; l < nl; ++l) { v[o++] = 2 * x[PF + l]; v[o++] = 2 * x[QF + l]; }
    for (int l = 0; l < nl; ++l) { v[o++] = 2 * x[PT + l]; v[o++] = 2 * x[QT + l]; }
    for (int k = 0; k < 4; ++k)
        for (int l = 0; l < nl; ++l) {
            const int fb = P->f_bus[l], tb = P->t_bus[l];
            double vrf = x[VR + fb], vif = x[VI + fb], vrt = x[VR + tb], vit = x[VI + tb];
            double A, Bc, Bs; int st; ohm_coef(P, l, k, &A, &Bc, &Bs, &st);
            v[o++] = 1.0;
            v[o++] = -((st ? 0.0 : 2 * A * vif) + Bc * vit + Bs * vrt);      /* d/d vi_f */
            v[o++] = -((st ? 2 * A * vit : 0.0) + Bc * vif - Bs * vrf);      /* d/d vi_t */
            v[o++] = -((st ? 0.0 : 2 * A * vrf) + Bc * vrt - Bs * vit);      /* d/d vr_f */
            v[o++] = -((st ? 2 * A * vrt : 0.0) + Bc * vrf + Bs * vif);      /* d/d vr_t */
        }
    for (int s = 0; s < P->nsh; ++s) {
        int i = P->sh_bus[s];
        v[o++] = 2 * P->sh_gs[s] * x[VR + i]; v[o++] = 2 * P->sh_gs[s] * x[VI + i];
        v[o++] = -2 * P->sh_bs[s] * x[VR + i]; v[o++] = -2 * P->sh_bs[s] * x[VI + i];
    }
    for (int d = 0; d < P->ndc; ++d) { v[o++] = 1.0 - P->dc_loss1[d]; v[o++] = 1.0; }
}
static void acr_h(void *u, const double *x, double sig, const double *lam, double *v)
{
    const ora_problem *P = (const ora_problem *)u; RIDX(P)
    (void)x;
    int64_t o = 0;
    for (int g = 0; g < ng; ++g) v[o++] = sig * 2 * P->c2[g];
    for (int l = 0; l < nl; ++l) { double w = 2 * lam[T0 + 2 * l]; v[o++] = w; v[o++] = w; }
    for (int l = 0; l < nl; ++l) { double w = 2 * lam[T0 + 2 * l + 1]; v[o++] = w; v[o++] = w; }
    for (int i = 0; i < nb; ++i) {
        double wl = 2 * lam[V0 + 2 * i], wu = 2 * lam[V0 + 2 * i + 1];
        v[o++] = wl; v[o++] = wl; v[o++] = wu; v[o++] = wu;
    }
    for (int k = 0; k < 4; ++k) {
        double *blk = v + o + (int64_t)k * 6 * nl;
        for (int l = 0; l < nl; ++l) {
            double A, Bc, Bs; int st; ohm_coef(P, l, k, &A, &Bc, &Bs, &st);
            double w = -lam[O0 + 4 * l + k];
            blk[0 * nl + l] = w * 2 * A;      /* vi_s vi_s */
            blk[1 * nl + l] = w * 2 * A;      /* vr_s vr_s */
            blk[2 * nl + l] = w * Bc;         /* vi_f vi_t */
            blk[3 * nl + l] = w * Bc;         /* vr_f vr_t */
            blk[4 * nl + l] = w * Bs;         /* vr_t vi_f */
            blk[5 * nl + l] = -w * Bs;        /* vr_f vi_t */
        }
    }
    o += (int64_t)24 * nl;
    for (int s = 0; s < P->nsh; ++s) {
        int i = P->sh_bus[s];
        double w = lam[1 + 2 * i] * 2 * P->sh_gs[s] - lam[2 + 2 * i] * 2 * P->sh_bs[s];
        v[o++] = w; v[o++] = w;
    }
}

ora_problem *ora_problem_acopf_acr(int nb, int ng, int nl, const int32_t *f_bus,
                                   const int32_t *t_bus, const double *ohm,
                                   const int32_t *gen_bus, const double *c2,
                                   const double *c1, const int32_t *bal_ptr, const int32_t *bal_colP,
                                   const int32_t *bal_colQ, const double *bal_coef,
                                   int64_t nnzj, const int64_t *jrow, const int64_t *jcol,
                                   int64_t nnzh, const int64_t *hrow, const int64_t *hcol,
                                   const double *xL, const double *xU, const double *gL,
                                   const double *gU, int nsh, const int32_t *sh_bus, const double *sh_gs,
                                   const double *sh_bs, int ndc, const double *dc_loss1)
{
    int64_t n = 2 * nb + 2 * ng + 4 * nl + 4 * ndc, m = 1 + 4 * nb + 6 * nl + ndc;
    double *x0 = (double *)calloc((size_t)n, sizeof(double));
    ora_problem *P = mk(n, m, nsh > 0 ? 1 : 1 + 2 * nb, nnzj, jrow, jcol, nnzh, hrow, hcol, xL, xU, gL, gU, x0);
    free(x0);
    P->nb = nb; P->ng = ng; P->nl = nl;
    P->ref_bus = (int)(jcol[0] - 1);
    P->f_bus = i32dup(f_bus, nl); P->t_bus = i32dup(t_bus, nl); P->gen_bus = i32dup(gen_bus, ng);
    P->bal_ptr = i32dup(bal_ptr, nb + 1);
    int64_t nbal = bal_ptr[nb];
    P->bal_colP = i32dup(bal_colP, nbal); P->bal_colQ = i32dup(bal_colQ, nbal);
    P->bal_coef = ddup(bal_coef, nbal);
    P->ohm = ddup(ohm, 12 * (int64_t)nl);
    P->ndc = ndc; P->dc_loss1 = ddup(dc_loss1, ndc);
    P->nsh = nsh; P->sh_bus = i32dup(sh_bus, nsh); P->sh_gs = ddup(sh_gs, nsh); P->sh_bs = ddup(sh_bs, nsh);
    P->c2 = ddup(c2, ng); P->c1 = ddup(c1, ng);
    P->nlp.eval_f = ac_f; P->nlp.eval_grad_f = ac_df; P->nlp.eval_g = acr_g;
    P->nlp.eval_jac_g = acr_jac; P->nlp.eval_h = acr_h;
    return P;
}

/* ------------------------------------------------------------------ ACOPF, W-space form (ACWR) */
/* /root/reference/examples/acopf/acwr.jl:1-37 over PowerModels' build_opf, laid out by sqpsolver.jl_amd/acopf_synth.py
 * acwr_layout: x = (vi, vr, w, wr, wi, pg, qg, flows, dc); every constraint of the W-R model is linear in (w, wr, wi)
 * and constraint_model_voltage ties them to (vr, vi) by nb + 2 nbp quadratic equalities. */
#define WIDX(P) \
    const int nb = (P)->nb, ng = (P)->ng, nl = (P)->nl, nbp = (P)->nbp; \
    const int VI = 0, VR = nb, W = 2 * nb, WR = 3 * nb, WI = 3 * nb + nbp, PG = 3 * nb + 2 * nbp, PF = PG + 2 * ng; \
    const int PT = PF + nl, QF = PF + 2 * nl, QT = PF + 3 * nl; \
    const int A0 = 1 + 2 * nb, O0 = A0 + 2 * nbp, V0 = O0 + 4 * nl, T0 = V0 + nb + 2 * nbp, D0 = T0 + 2 * nl; \
    (void)VI; (void)VR; (void)W; (void)WR; (void)WI; (void)PG; (void)PT; (void)QF; (void)QT; (void)A0; (void)O0; (void)V0; \
    (void)T0; (void)D0; (void)ng;

static double wr_f(void *u, const double *x)
{
    const ora_problem *P = (const ora_problem *)u; WIDX(P)
    double f = 0.0;
    for (int g = 0; g < ng; ++g) f += P->c2[g] * x[PG + g] * x[PG + g] + P->c1[g] * x[PG + g];
    return f;
}
static void wr_df(void *u, const double *x, double *gr)
{
    const ora_problem *P = (const ora_problem *)u; WIDX(P)
    memset(gr, 0, sizeof(double) * (size_t)P->nlp.n);
    for (int g = 0; g < ng; ++g) gr[PG + g] = 2 * P->c2[g] * x[PG + g] + P->c1[g];
}
static void wr_g(void *u, const double *x, double *gv)
{
    const ora_problem *P = (const ora_problem *)u; WIDX(P)
    const int own[4] = { PF, QF, PT, QT };
    gv[0] = x[VI + P->ref_bus];
    for (int i = 0; i < nb; ++i) {
        double sp = 0.0, sq = 0.0;
        for (int k = P->bal_ptr[i]; k < P->bal_ptr[i + 1]; ++k) {
            sp += P->bal_coef[k] * x[P->bal_colP[k]];
            sq += P->bal_coef[k] * x[P->bal_colQ[k]];
        }
        gv[1 + 2 * i] = sp + P->gsb[i] * x[W + i]; gv[2 + 2 * i] = sq - P->bsb[i] * x[W + i];
        gv[V0 + i] = x[W + i] - x[VR + i] * x[VR + i] - x[VI + i] * x[VI + i];
    }
    for (int k = 0; k < nbp; ++k) {
        const int i = P->bp_i[k], j = P->bp_j[k];
        gv[A0 + 2 * k] = x[WI + k] - P->bp_tmax[k] * x[WR + k];
        gv[A0 + 2 * k + 1] = x[WI + k] - P->bp_tmin[k] * x[WR + k];
        gv[V0 + nb + 2 * k] = x[WR + k] - (x[VR + i] * x[VR + j] + x[VI + i] * x[VI + j]);
        gv[V0 + nb + 2 * k + 1] = x[WI + k] - (x[VI + i] * x[VR + j] - x[VR + i] * x[VI + j]);
    }
    for (int l = 0; l < nl; ++l) {
        gv[T0 + 2 * l] = x[PF + l] * x[PF + l] + x[QF + l] * x[QF + l];
        gv[T0 + 2 * l + 1] = x[PT + l] * x[PT + l] + x[QT + l] * x[QT + l];
        const int k = P->br_bp[l]; const double sg = P->br_sig[l];
        for (int c = 0; c < 4; ++c) {
            double A, Bc, Bs; int st; ohm_coef(P, l, c, &A, &Bc, &Bs, &st);
            const double ws = x[W + (st ? P->t_bus[l] : P->f_bus[l])];
            gv[O0 + 4 * l + c] = x[own[c] + l] - (A * ws + Bc * x[WR + k] + sg * Bs * x[WI + k]);
        }
    }
    for (int d = 0; d < P->ndc; ++d)
        gv[D0 + d] = (1.0 - P->dc_loss1[d]) * x[PF + 4 * nl + d] + x[PF + 4 * nl + P->ndc + d];
}
static void wr_jac(void *u, const double *x, double *v)
{
    const ora_problem *P = (const ora_problem *)u; WIDX(P)
    int64_t o = 0;
    v[o++] = 1.0;
    for (int i = 0; i < nb; ++i) {
        int s = P->bal_ptr[i], e = P->bal_ptr[i + 1];
        for (int k = s; k < e; ++k) v[o++] = P->bal_coef[k];
        for (int k = s; k < e; ++k) v[o++] = P->bal_coef[k];
    }
    for (int i = 0; i < nb; ++i) { v[o++] = P->gsb[i]; v[o++] = -P->bsb[i]; }
    for (int k = 0; k < nbp; ++k) { v[o++] = 1.0; v[o++] = -P->bp_tmax[k]; }
    for (int k = 0; k < nbp; ++k) { v[o++] = 1.0; v[o++] = -P->bp_tmin[k]; }
    for (int c = 0; c < 4; ++c)
        for (int l = 0; l < nl; ++l) {
            double A, Bc, Bs; int st; ohm_coef(P, l, c, &A, &Bc, &Bs, &st);
            v[o++] = 1.0; v[o++] = -A; v[o++] = -Bc; v[o++] = -P->br_sig[l] * Bs;
        }
    for (int i = 0; i < nb; ++i) { v[o++] = 1.0; v[o++] = -2 * x[VR + i]; v[o++] = -2 * x[VI + i]; }
    for (int k = 0; k < nbp; ++k) {
        const int i = P->bp_i[k], j = P->bp_j[k];
        v[o++] = 1.0; v[o++] = -x[VR + j]; v[o++] = -x[VR + i]; v[o++] = -x[VI + j]; v[o++] = -x[VI + i];
    }
    for (int k = 0; k < nbp; ++k) {
        const int i = P->bp_i[k], j = P->bp_j[k];
        v[o++] = 1.0; v[o++] = -x[VR + j]; v[o++] = -x[VI + i]; v[o++] = x[VI + j]; v[o++] = x[VR + i];
    }
    for (int l = 0; l < nl; ++l) { v[o++] = 2 * x[PF + l]; v[o++] = 2 * x[QF + l]; }
    for (int l = 0; l < nl; ++l) { v[o++] = 2 * x[PT + l]; v[o++] = 2 * x[QT + l]; }
    for (int d = 0; d < P->ndc; ++d) { v[o++] = 1.0 - P->dc_loss1[d]; v[o++] = 1.0; }
}
static void wr_h(void *u, const double *x, double sig, const double *lam, double *v)
{
    const ora_problem *P = (const ora_problem *)u; WIDX(P)
    (void)x;
    int64_t o = 0;
    for (int g = 0; g < ng; ++g) v[o++] = sig * 2 * P->c2[g];
    for (int l = 0; l < nl; ++l) { double w = 2 * lam[T0 + 2 * l]; v[o++] = w; v[o++] = w; }
    for (int l = 0; l < nl; ++l) { double w = 2 * lam[T0 + 2 * l + 1]; v[o++] = w; v[o++] = w; }
    for (int i = 0; i < nb; ++i) { double w = -2 * lam[V0 + i]; v[o++] = w; v[o++] = w; }
    for (int k = 0; k < nbp; ++k) { double w = -lam[V0 + nb + 2 * k]; v[o++] = w; v[o++] = w; }
    for (int k = 0; k < nbp; ++k) { double w = lam[V0 + nb + 2 * k + 1]; v[o++] = -w; v[o++] = w; }
}

ora_problem *ora_problem_acopf_acwr(int nb, int ng, int nl, const int32_t *f_bus,
                                    const int32_t *t_bus, const double *ohm,
                                    const int32_t *gen_bus, const double *c2,
                                    const double *c1, const int32_t *bal_ptr, const int32_t *bal_colP,
                                    const int32_t *bal_colQ, const double *bal_coef,
                                    int64_t nnzj, const int64_t *jrow, const int64_t *jcol,
                                    int64_t nnzh, const int64_t *hrow, const int64_t *hcol,
                                    const double *xL, const double *xU, const double *gL,
                                    const double *gU, int nsh, const int32_t *sh_bus, const double *sh_gs,
                                    const double *sh_bs, int ndc, const double *dc_loss1,
                                    int nbp, const int32_t *bp_i, const int32_t *bp_j, const int32_t *br_bp,
                                    const double *br_sig, const double *bp_tmin, const double *bp_tmax)
{
    int64_t n = 3 * nb + 2 * nbp + 2 * ng + 4 * nl + 4 * ndc;
    int64_t m = 1 + 2 * nb + 2 * nbp + 4 * nl + nb + 2 * nbp + 2 * nl + ndc;
    double *x0 = (double *)calloc((size_t)n, sizeof(double));
    ora_problem *P = mk(n, m, 1 + 2 * nb + 2 * nbp + 4 * nl, nnzj, jrow, jcol, nnzh, hrow, hcol, xL, xU, gL, gU, x0);
    free(x0);
    P->nb = nb; P->ng = ng; P->nl = nl; P->nbp = nbp;
    P->ref_bus = (int)(jcol[0] - 1);
    P->f_bus = i32dup(f_bus, nl); P->t_bus = i32dup(t_bus, nl); P->gen_bus = i32dup(gen_bus, ng);
    P->bal_ptr = i32dup(bal_ptr, nb + 1);
    int64_t nbal = bal_ptr[nb];
    P->bal_colP = i32dup(bal_colP, nbal); P->bal_colQ = i32dup(bal_colQ, nbal);
    P->bal_coef = ddup(bal_coef, nbal);
    P->ohm = ddup(ohm, 12 * (int64_t)nl);
    P->ndc = ndc; P->dc_loss1 = ddup(dc_loss1, ndc);
    P->nsh = nsh; P->sh_bus = i32dup(sh_bus, nsh); P->sh_gs = ddup(sh_gs, nsh); P->sh_bs = ddup(sh_bs, nsh);
    P->gsb = (double *)calloc((size_t)nb, sizeof(double)); P->bsb = (double *)calloc((size_t)nb, sizeof(double));
    for (int s = 0; s < nsh; ++s) { P->gsb[sh_bus[s]] = sh_gs[s]; P->bsb[sh_bus[s]] = sh_bs[s]; }
    P->bp_i = i32dup(bp_i, nbp); P->bp_j = i32dup(bp_j, nbp); P->br_bp = i32dup(br_bp, nl);
    P->br_sig = ddup(br_sig, nl); P->bp_tmin = ddup(bp_tmin, nbp); P->bp_tmax = ddup(bp_tmax, nbp);
    P->c2 = ddup(c2, ng); P->c1 = ddup(c1, ng);
    P->nlp.eval_f = wr_f; P->nlp.eval_grad_f = wr_df; P->nlp.eval_g = wr_g;
    P->nlp.eval_jac_g = wr_jac; P->nlp.eval_h = wr_h;
    return P;
}


/* ------------------------------------------------------------------ synthetic dense-Hessian NLP
 * The CPU twin of dense_eval (sqpsolver.jl_amd/csrc/acopf_dev.hpp), bench.py --workload dense:
 *     min 1/2 x'Qx + c'x + kappa/4 sum x_i^4   s.t.  A x = b,  xL <= x <= xU
 * No reference counterpart (the reference's examples are ACOPF models); the callbacks have the shape of
 * SqpSolver.Model (src/model.jl:3-35).  Jacobian COO = A row-major, Hessian COO = lower triangle of Q column-major. */
static double dn_f(void *u, const double *x)
{
    const ora_problem *P = (const ora_problem *)u;
    const int64_t n = P->nlp.n;
    double f = 0.0;
    for (int64_t j = 0; j < n; ++j) {
        double acc = 0.0;
        for (int64_t i = 0; i < n; ++i) acc += P->dnQ[i * n + j] * x[i];
        const double x2 = x[j] * x[j];
        f += 0.5 * x[j] * acc + P->dnc[j] * x[j] + 0.25 * P->dn_kappa * x2 * x2;
    }
    return f;
}
static void dn_df(void *u, const double *x, double *g)
{
    const ora_problem *P = (const ora_problem *)u;
    const int64_t n = P->nlp.n;
    for (int64_t j = 0; j < n; ++j) {
        double acc = 0.0;
        for (int64_t i = 0; i < n; ++i) acc += P->dnQ[i * n + j] * x[i];
        g[j] = acc + P->dnc[j] + P->dn_kappa * (x[j] * x[j]) * x[j];
    }
}
static void dn_g(void *u, const double *x, double *g)
{
    const ora_problem *P = (const ora_problem *)u;
    const int64_t n = P->nlp.n, m = P->nlp.m;
    for (int64_t i = 0; i < m; ++i) {
        double acc = 0.0;
        for (int64_t j = 0; j < n; ++j) acc += P->dnA[i * n + j] * x[j];
        g[i] = acc;
    }
}
static void dn_jac(void *u, const double *x, double *v)
{
    const ora_problem *P = (const ora_problem *)u; (void)x;
    memcpy(v, P->dnA, sizeof(double) * (size_t)(P->nlp.n * P->nlp.m));
}
static void dn_h(void *u, const double *x, double s, const double *l, double *v)
{
    const ora_problem *P = (const ora_problem *)u; (void)l;
    const int64_t n = P->nlp.n;
    for (int64_t j = 0; j < n; ++j) {
        const int64_t off = j * n - j * (j - 1) / 2;
        for (int64_t i = j; i < n; ++i)
            v[off + i - j] = s * (P->dnQ[j * n + i] + (i == j ? 3.0 * P->dn_kappa * x[j] * x[j] : 0.0));
    }
}
ora_problem *ora_problem_dense(int64_t n, int64_t m, const double *Q, const double *A, const double *c, double kappa,
                               int64_t nnzj, const int64_t *jrow, const int64_t *jcol, int64_t nnzh, const int64_t *hrow,
                               const int64_t *hcol, const double *xL, const double *xU, const double *gL, const double *gU,
                               const double *x0)
{
    if (nnzj != n * m || nnzh != n * (n + 1) / 2) return NULL;
    ora_problem *P = mk(n, m, m, nnzj, jrow, jcol, nnzh, hrow, hcol, xL, xU, gL, gU, x0);
    P->dnQ = ddup(Q, n * n); P->dnA = ddup(A, m * n); P->dnc = ddup(c, n); P->dn_kappa = kappa;
    P->nlp.eval_f = dn_f; P->nlp.eval_grad_f = dn_df; P->nlp.eval_g = dn_g; P->nlp.eval_jac_g = dn_jac; P->nlp.eval_h = dn_h;
    return P;
}
