/* sqp_tr.c -- literal CPU restatement of SqpTR / run!.  TEST INFRASTRUCTURE ONLY.
 * Follows /root/reference/src/algorithms/sqp_trust_region.jl and sqp.jl line by line
 * (citations at each step).  Quirks of the reference are reproduced on purpose and are
 * catalogued in SURVEY.md Appendix C. */
#include "sqp_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef struct {
    const ora_nlp *nlp;
    ora_options opt;
    int64_t n, m;
    /* sqp.jl:16-59 fields */
    double *x, *p, *p_soc, *lambda, *mult_x_L, *mult_x_U;
    double f, *df, *E, *dE, *h_val;
    int64_t *jcolptr, *jrowval, *jslot;          /* Jacobian CSC + COO->slot map */
    double *jnz;
    int64_t *hcolptr, *hrowval, *hslot, *hslot_t; /* Hessian CSC (both triangles) */
    double *hnz;
    int64_t nnzjc, nnzhc;
    double prim_infeas, dual_infeas;
    ora_qp *optimizer;
    int sub_status;
    int feasibility_restoration, iter, ret;
    double *tmpx, *tmpE;
    /* sqp_trust_region.jl:6-24 */
    double *p_lambda, *p_mult_x_L, *p_mult_x_U, *E_soc;
    double phi, mu, Delta, Delta_max;
    int step_acceptance;
    /* bookkeeping */
    double *q_lambda, *q_mxU, *q_mxL, *q_p;
    ora_result *res;
    ora_trace_row *trace; int trace_cap;
    int it_ipm, it_fac;
} sqp_t;

static double now_s(void)
{
    struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
static double *dal(int64_t k) { return (double *)calloc((size_t)(k > 0 ? k : 1), sizeof(double)); }
static double norm_inf(const double *v, int64_t k)
{ double a = 0.0; for (int64_t i = 0; i < k; ++i) a = fmax(a, fabs(v[i])); return a; }

/* COO (1-based, duplicates) -> CSC pattern with duplicates merged, as Julia's sparse(I,J,V,m,n)
 * (sqp_trust_region.jl:47-48,56-57). If sym, both (r,c) and (c,r) are inserted (sqp.jl:96-101). */
typedef struct { int64_t r, c, k, t; } ent_t;
static int ent_cmp(const void *a, const void *b)
{
    const ent_t *x = (const ent_t *)a, *y = (const ent_t *)b;
    if (x->c != y->c) return x->c < y->c ? -1 : 1;
    if (x->r != y->r) return x->r < y->r ? -1 : 1;
    return 0;
}
static int64_t build_csc(int64_t ncols, int64_t nnz, const int64_t *row, const int64_t *col, int sym,
                         int64_t **colptr, int64_t **rowval, int64_t **slot, int64_t **slot_t)
{
    int64_t cap = sym ? 2 * nnz : nnz;
    ent_t *e = (ent_t *)malloc(sizeof(ent_t) * (size_t)(cap + 1));
    int64_t ne = 0;
    for (int64_t k = 0; k < nnz; ++k) {
        e[ne++] = (ent_t){ row[k] - 1, col[k] - 1, k, 0 };
        if (sym && row[k] != col[k]) e[ne++] = (ent_t){ col[k] - 1, row[k] - 1, k, 1 };
    }
    qsort(e, (size_t)ne, sizeof(ent_t), ent_cmp);
    *colptr = (int64_t *)calloc((size_t)(ncols + 1), sizeof(int64_t));
    *rowval = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ne + 1));
    *slot = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nnz + 1));
    if (slot_t) { *slot_t = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nnz + 1)); for (int64_t k = 0; k < nnz; ++k) (*slot_t)[k] = -1; }
    int64_t ns = 0;
    for (int64_t i = 0; i < ne; ++i) {
        if (i == 0 || e[i].c != e[i - 1].c || e[i].r != e[i - 1].r) {
            (*rowval)[ns] = e[i].r;
            (*colptr)[e[i].c + 1]++;
            ++ns;
        }
        if (e[i].t == 0) (*slot)[e[i].k] = ns - 1; else (*slot_t)[e[i].k] = ns - 1;
    }
    for (int64_t j = 0; j < ncols; ++j) (*colptr)[j + 1] += (*colptr)[j];
    free(e);
    return ns;
}

/* sqp.jl:111-117 */
static void eval_jacobian(sqp_t *s)
{
    s->nlp->eval_jac_g(s->nlp->ud, s->x, s->dE);
    memset(s->jnz, 0, sizeof(double) * (size_t)s->nnzjc);
    for (int64_t k = 0; k < s->nlp->nnzj; ++k) s->jnz[s->jslot[k]] += s->dE[k];
}

/* sqp.jl:86-104 -- NB the Hessian is evaluated with sqp.lambda as is (JuMP sign), quirk #2 */
static void eval_functions(sqp_t *s)
{
    const ora_nlp *P = s->nlp;
    s->f = P->eval_f(P->ud, s->x);
    P->eval_grad_f(P->ud, s->x, s->df);
    P->eval_g(P->ud, s->x, s->E);
    eval_jacobian(s);
    if (P->eval_h && P->nnzh > 0) {
        if (s->opt.literal_quirks) P->eval_h(P->ud, s->x, 1.0, s->lambda, s->h_val);
        else {
            for (int64_t i = 0; i < s->m; ++i) s->tmpE[i] = -s->lambda[i];
            P->eval_h(P->ud, s->x, 1.0, s->tmpE, s->h_val);
        }
        memset(s->hnz, 0, sizeof(double) * (size_t)s->nnzhc);
        for (int64_t k = 0; k < P->nnzh; ++k) {
            s->hnz[s->hslot[k]] += s->h_val[k];
            if (s->hslot_t[k] >= 0) s->hnz[s->hslot_t[k]] += s->h_val[k];
        }
    }
}

static double viol1(sqp_t *s, const double *E, const double *x)
{ return ora_norm_violations(s->m, s->n, E, s->nlp->gL, s->nlp->gU, x, s->nlp->xL, s->nlp->xU, 1); }

/* sqp.jl:170-183 */
static double compute_phi(sqp_t *s, const double *x, double alpha, const double *p)
{
    for (int64_t j = 0; j < s->n; ++j) s->tmpx[j] = x[j] + alpha * p[j];
    double f = s->f;
    memcpy(s->tmpE, s->E, sizeof(double) * (size_t)s->m);
    if (alpha > 0.0) {
        f = s->nlp->eval_f(s->nlp->ud, s->tmpx);
        s->nlp->eval_g(s->nlp->ud, s->tmpx, s->tmpE);
    }
    if (s->feasibility_restoration) return viol1(s, s->tmpE, s->tmpx);
    return f + s->mu * viol1(s, s->tmpE, s->tmpx);
}

/* sqp_trust_region.jl:487-508 */
static double compute_qmodel(sqp_t *s, const double *p, int with_step)
{
    double q = 0.0;
    if (with_step) {
        double dfp = 0.0, php = 0.0;
        for (int64_t j = 0; j < s->n; ++j) dfp += s->df[j] * p[j];
        for (int64_t j = 0; j < s->n; ++j) {
            double acc = 0.0;   /* (H p)_... accumulated column-wise: p' H p */
            for (int64_t k = s->hcolptr[j]; k < s->hcolptr[j + 1]; ++k) acc += s->hnz[k] * p[s->hrowval[k]];
            php += acc * p[j];
        }
        q += dfp + 0.5 * php;
        for (int64_t j = 0; j < s->n; ++j) s->tmpx[j] = s->x[j] + p[j];
        memcpy(s->tmpE, s->E, sizeof(double) * (size_t)s->m);
        for (int64_t j = 0; j < s->n; ++j)
            for (int64_t k = s->jcolptr[j]; k < s->jcolptr[j + 1]; ++k)
                s->tmpE[s->jrowval[k]] += s->jnz[k] * p[j];
    } else {
        memcpy(s->tmpx, s->x, sizeof(double) * (size_t)s->n);
        memcpy(s->tmpE, s->E, sizeof(double) * (size_t)s->m);
    }
    return q + s->mu * viol1(s, s->tmpE, s->tmpx);
}

static void push_trace(sqp_t *s)
{
    if (!s->trace || s->res->trace_len >= s->trace_cap) { s->res->trace_len++; return; }
    ora_trace_row *r = &s->trace[s->res->trace_len++];
    r->iter = s->iter; r->accepted = s->step_acceptance; r->fr = s->feasibility_restoration;
    r->sub_status = s->sub_status; r->ipm_iters = s->it_ipm; r->n_factor = s->it_fac;
    r->f = s->f; r->phi = s->phi; r->mu = s->mu; r->delta = s->Delta;
    r->pnorm = norm_inf(s->p, s->n); r->prim_infeas = s->prim_infeas; r->dual_infeas = s->dual_infeas;
}

static int qp_call(sqp_t *s, int mode, const double *b, double *p, double *lam, double *mxU, double *mxL)
{
    double t0 = now_s();
    int st = ora_qp_solve(s->optimizer, mode, s->x, s->Delta, s->mu, s->df, b, s->jnz,
                          s->nnzhc ? s->hnz : NULL, p, lam, mxU, mxL, NULL);
    s->res->qp_seconds += now_s() - t0;
    int a, b2; ora_qp_stats(s->optimizer, &a, &b2, NULL);
    s->res->n_qp++; s->res->n_ipm_iter += a; s->res->n_factor += b2;
    s->it_ipm += a; s->it_fac += b2;
    if (getenv("ORA_QP_LOG")) {          /* experiment aid: one line per sub-problem (mode, status, work, radius, last acceptance) */
        FILE *fh = fopen(getenv("ORA_QP_LOG"), "a");
        int rule; double e0;
        ora_qp_termination(s->optimizer, &rule, &e0);
        if (fh) { fprintf(fh, "%d %d %d %d %d %.6e %d %d %.6e\n", mode, st, a, b2, s->iter, s->Delta, s->step_acceptance, rule, e0); fclose(fh); }
    }
    return st;
}

/* sqp_trust_region.jl:370-380 with :314-331 inlined */
static void compute_step(sqp_t *s)
{
    int mode = s->feasibility_restoration ? ORA_MODE_FR : ORA_MODE_QP;
    s->sub_status = qp_call(s, mode, s->E, s->p, s->q_lambda, s->q_mxU, s->q_mxL);
    for (int64_t i = 0; i < s->m; ++i) s->p_lambda[i] = s->q_lambda[i] - s->lambda[i];
    for (int64_t j = 0; j < s->n; ++j) {
        s->p_mult_x_L[j] = s->q_mxL[j] - s->mult_x_L[j];
        s->p_mult_x_U[j] = s->q_mxU[j] - s->mult_x_U[j];
    }
    /* :378 -- uses the CURRENT multipliers (quirk #5) */
    s->mu = fmax(fmax(s->mu, norm_inf(s->lambda, s->m)),
                 fmax(norm_inf(s->mult_x_L, s->n), norm_inf(s->mult_x_U, s->n)));
}

/* sqp_trust_region.jl:341-360 */
static void sub_optimize_soc(sqp_t *s)
{
    for (int64_t j = 0; j < s->n; ++j) s->tmpx[j] = s->x[j] + s->p[j];
    s->nlp->eval_g(s->nlp->ud, s->tmpx, s->E_soc);
    for (int64_t j = 0; j < s->n; ++j)
        for (int64_t k = s->jcolptr[j]; k < s->jcolptr[j + 1]; ++k)
            s->E_soc[s->jrowval[k]] -= s->jnz[k] * s->p[j];
    double *lam = dal(s->m), *u = dal(s->n), *l = dal(s->n);
    qp_call(s, ORA_MODE_SOC, s->E_soc, s->q_p, lam, u, l);
    for (int64_t j = 0; j < s->n; ++j) s->p_soc[j] = s->p[j] + s->q_p[j];
    free(lam); free(u); free(l);
}

static void accept(sqp_t *s, const double *step)
{
    for (int64_t j = 0; j < s->n; ++j) {
        s->x[j] += step[j];
        s->mult_x_L[j] += s->p_mult_x_L[j];
        s->mult_x_U[j] += s->p_mult_x_U[j];
    }
    for (int64_t i = 0; i < s->m; ++i) s->lambda[i] += s->p_lambda[i];
}

/* sqp_trust_region.jl:515-579 */
static void do_step(sqp_t *s)
{
    double phi_k = compute_phi(s, s->x, 1.0, s->p);
    double ared = s->phi - phi_k;
    double pred = 1.0, q_0 = 0.0;
    if (!s->feasibility_restoration) {
        q_0 = compute_qmodel(s, s->p, 0);
        double q_k = compute_qmodel(s, s->p, 1);
        pred = q_0 - q_k;
    }
    double rho = ared / pred;
    double pn = norm_inf(s->p, s->n);
    if (ared > 0 && rho > 0) {
        accept(s, s->p);
        if (ora_isapprox(s->Delta, pn)) s->Delta = fmin(2 * s->Delta, s->Delta_max);
        s->step_acceptance = 1;
    } else {
        for (int64_t j = 0; j < s->n; ++j) s->tmpx[j] = s->x[j] + s->p[j];
        memset(s->tmpE, 0, sizeof(double) * (size_t)s->m);
        s->nlp->eval_g(s->nlp->ud, s->tmpx, s->tmpE);
        double c_k = viol1(s, s->tmpE, s->tmpx);
        int perform_soc = 0;
        if (s->opt.use_soc && c_k > 0 && !s->feasibility_restoration) {
            sub_optimize_soc(s);
            double phi_soc = compute_phi(s, s->x, 1.0, s->p_soc);
            ared = s->phi - phi_soc;
            double q_soc = compute_qmodel(s, s->p_soc, 1);
            pred = q_0 - q_soc;
            double rho_soc = ared / pred;
            if (ared > 0 && rho_soc > 0) {
                accept(s, s->p_soc);
                s->step_acceptance = 1;
                perform_soc = 1;
            }
        }
        if (!perform_soc) {
            s->Delta = fmax(0.5 * fmin(s->Delta, pn), 0.1 * s->opt.tol_direction);
            s->step_acceptance = 0;
        }
    }
}

/* utils.jl:16-22 */
static void dropzeros(double *v, int64_t k)
{ for (int64_t i = 0; i < k; ++i) if (fabs(v[i]) < 1e-10) v[i] = 0.0; }

void ora_sqp_tr_solve(const ora_nlp *nlp, const ora_options *opt, double *x, double *g,
                      double *mult_g, double *mult_x_L, double *mult_x_U,
                      ora_result *res, ora_trace_row *trace, int trace_cap)
{
    sqp_t S; memset(&S, 0, sizeof(S));
    sqp_t *s = &S;
    int64_t n = nlp->n, m = nlp->m;
    s->nlp = nlp; s->opt = *opt; s->n = n; s->m = m;
    s->res = res; s->trace = trace; s->trace_cap = trace_cap;
    memset(res, 0, sizeof(*res));
    /* SqpTR constructor, sqp_trust_region.jl:26-91 */
    s->x = dal(n); memcpy(s->x, x, sizeof(double) * (size_t)n);
    s->p = dal(n); s->p_soc = dal(n); s->lambda = dal(m); s->mult_x_L = dal(n); s->mult_x_U = dal(n);
    s->df = dal(n); s->E = dal(m); s->dE = dal(nlp->nnzj); s->h_val = dal(nlp->nnzh);
    s->nnzjc = build_csc(n, nlp->nnzj, nlp->jrow, nlp->jcol, 0, &s->jcolptr, &s->jrowval, &s->jslot, NULL);
    s->jnz = dal(s->nnzjc);
    s->nnzhc = build_csc(n, nlp->nnzh, nlp->hrow, nlp->hcol, 1, &s->hcolptr, &s->hrowval, &s->hslot, &s->hslot_t);
    s->hnz = dal(s->nnzhc);
    s->p_lambda = dal(m); s->p_mult_x_L = dal(n); s->p_mult_x_U = dal(n); s->E_soc = dal(m);
    s->tmpx = dal(n); s->tmpE = dal(m);
    s->q_lambda = dal(m); s->q_mxU = dal(n); s->q_mxL = dal(n); s->q_p = dal(n);
    s->phi = 1e20; s->Delta_max = 1e8; s->step_acceptance = 1;
    s->prim_infeas = INFINITY; s->dual_infeas = INFINITY;
    s->feasibility_restoration = 0; s->iter = 1; s->ret = -5;
    s->optimizer = ora_qp_create(n, m, nlp->num_linear, s->jcolptr, s->jrowval, s->hcolptr,
                                 s->hrowval, nlp->xL, nlp->xU, nlp->gL, nlp->gU, opt);

    /* run!, sqp_trust_region.jl:98-223 */
    s->mu = opt->init_mu;
    s->Delta = opt->tr_size;
    int early_nan = 0;
    {   /* :237-254 */
        s->f = nlp->eval_f(nlp->ud, s->x);
        if (!isnan(s->f)) nlp->eval_g(nlp->ud, s->x, s->E);
        double lpviol = 0.0;
        for (int64_t i = 0; i < nlp->num_linear; ++i) {
            lpviol += fmax(0.0, nlp->gL[i] - s->E[i]);
            lpviol -= fmin(0.0, nlp->gU[i] - s->E[i]);
        }
        for (int64_t j = 0; j < n; ++j) {
            lpviol += fmax(0.0, nlp->xL[j] - s->x[j]);
            lpviol -= fmin(0.0, nlp->xU[j] - s->x[j]);
        }
        if (isnan(s->f)) { early_nan = 1; }
        else if (lpviol > opt->tol_infeas) {
            /* sub_optimize_lp!, :264-304 */
            s->f = nlp->eval_f(nlp->ud, s->x);
            nlp->eval_grad_f(nlp->ud, s->x, s->df);
            eval_jacobian(s);
            double *xs = dal(n);
            s->it_ipm = s->it_fac = 0;
            s->sub_status = qp_call(s, ORA_MODE_LP, NULL, xs, s->lambda, s->mult_x_U, s->mult_x_L);
            memcpy(s->x, xs, sizeof(double) * (size_t)n);
            free(xs);
            dropzeros(s->x, n); dropzeros(s->lambda, m);
            dropzeros(s->mult_x_U, n); dropzeros(s->mult_x_L, n);
            push_trace(s);   /* print(sqp, "LP") */
        }
    }
    if (early_nan) {
        res->status = -13;   /* :113-115, returns before any write-back */
        goto done;
    }

    for (;;) {
        /* sqp.jl:215-224 */
        if (s->iter > opt->max_iter) {
            s->ret = -1;
            if (s->prim_infeas <= opt->tol_infeas) s->ret = 6;
            break;
        }
        s->it_ipm = s->it_fac = 0;
        if (s->step_acceptance) {           /* :134-138 */
            eval_functions(s);
            s->prim_infeas = viol1(s, s->E, s->x);
            if (opt->literal_quirks)
                s->dual_infeas = ora_kt_residuals(m, n, s->df, s->lambda, s->mult_x_U, s->mult_x_L,
                                                  s->jcolptr, s->jrowval, s->jnz);
            else {   /* df - J'lambda - mult_x_L - mult_x_U, same scaling */
                for (int64_t i = 0; i < m; ++i) s->tmpE[i] = -s->lambda[i];
                for (int64_t j = 0; j < n; ++j) s->tmpx[j] = -s->mult_x_U[j];
                s->dual_infeas = ora_kt_residuals(m, n, s->df, s->tmpE, s->tmpx, s->mult_x_L,
                                                  s->jcolptr, s->jrowval, s->jnz);
            }
        }
        compute_step(s);                    /* :141 */
        int st = s->sub_status;
        if (st == ORA_MOI_OPTIMAL || st == ORA_MOI_ALMOST_OPTIMAL ||
            st == ORA_MOI_ALMOST_LOCALLY_SOLVED || st == ORA_MOI_LOCALLY_SOLVED) {
            if (s->Delta == s->Delta_max && ora_isapprox(norm_inf(s->p, n), s->Delta)) {
                s->ret = 4;                 /* :146-150 */
                break;
            }
        } else if (st == ORA_MOI_INFEASIBLE || st == ORA_MOI_LOCALLY_INFEASIBLE) {
            if (s->feasibility_restoration) {   /* :152-159 */
                s->ret = s->prim_infeas <= opt->tol_infeas ? 6 : 2;
                break;
            } else {                            /* :160-168 */
                s->feasibility_restoration = 1;
                push_trace(s);
                s->iter += 1;
                continue;
            }
        } else {                                /* :169-178; `sqp.ret == -3` is a no-op (quirk #1) */
            if (s->prim_infeas <= opt->tol_infeas * 10.0) s->ret = 6;
            break;
        }
        if (s->step_acceptance) s->phi = compute_phi(s, s->x, 0.0, s->p);   /* :180-182 */
        push_trace(s);                          /* :184 */
        double pn = norm_inf(s->p, n);
        if (pn <= opt->tol_direction) {         /* :187-196 */
            if (s->feasibility_restoration) {
                s->feasibility_restoration = 0;
                s->iter += 1;
                continue;
            } else { s->ret = 0; break; }
        }
        if (s->prim_infeas <= opt->tol_infeas && s->dual_infeas <= opt->tol_residual &&
            !ora_isapprox(s->Delta, pn) && !s->feasibility_restoration) {   /* :198-204 */
            s->ret = 0;
            break;
        }
        do_step(s);                             /* :206 */
        if (s->feasibility_restoration && s->step_acceptance) s->feasibility_restoration = 0; /* :209-211 */
        s->iter += 1;
    }
    /* :215-222 */
    res->obj_val = nlp->eval_f(nlp->ud, s->x);
    res->status = s->ret;
    memcpy(x, s->x, sizeof(double) * (size_t)n);
    memcpy(g, s->E, sizeof(double) * (size_t)m);
    for (int64_t i = 0; i < m; ++i) mult_g[i] = -s->lambda[i];
    for (int64_t j = 0; j < n; ++j) { mult_x_U[j] = -s->mult_x_U[j]; mult_x_L[j] = s->mult_x_L[j]; }
done:
    res->iter = s->iter;
    ora_qp_destroy(s->optimizer);
    void *ptrs[] = { s->x, s->p, s->p_soc, s->lambda, s->mult_x_L, s->mult_x_U, s->df, s->E, s->dE,
        s->h_val, s->jcolptr, s->jrowval, s->jslot, s->jnz, s->hcolptr, s->hrowval, s->hslot,
        s->hslot_t, s->hnz, s->p_lambda, s->p_mult_x_L, s->p_mult_x_U, s->E_soc, s->tmpx, s->tmpE,
        s->q_lambda, s->q_mxU, s->q_mxL, s->q_p };
    for (size_t i = 0; i < sizeof(ptrs) / sizeof(ptrs[0]); ++i) free(ptrs[i]);
}
