/* qp_ipm.c -- the QP sub-problem of SqpSolver.jl on the CPU.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference builds a JuMP model of the trust-region QP and hands it to an external solver
 * (/root/reference/src/algorithms/subproblem_JuMP.jl:127-183; Ipopt in every test and example).
 * Ipopt is not vendored and cannot run here, so this file is NOT a restatement of Ipopt: it is a
 * primal-dual interior-point method written for this project that solves the same mathematical
 * programmes and reports results the way collect_solution! does
 * (subproblem_JuMP.jl:514-563).  "parity unpinned" for the arithmetic; pinned by the reference's
 * known answers and KKT checks in tests/.
 *
 * Canonical programme (every mode maps onto it, see setup_mode):
 *     min  c'p + 1/2 p'(H + diag(hd))p + sum_i (wp_i tp_i + wm_i tm_i)
 *     s.t. J_i p + tp_i - tm_i - s_i = 0,  lo_i <= s_i <= hi_i      (inequality / ranged rows)
 *          J_i p + tp_i - tm_i       = lo_i                          (equality rows)
 *          tp, tm >= 0,   lb <= p <= ub
 * Rows the mode treats as hard carry the exact-penalty weight rho_big; if elastic mass remains on a
 * hard row the sub-problem is reported infeasible (with opt.ipm_phase1 a phase-1 solve first decides
 * between "infeasible" and "raise rho_big"; on every problem tried it only ever confirmed the verdict).  With elastics on every
 * row the reduced KKT matrix  K = [W J'; J -D]  (W = H + hd + Sigma_p + delta_w I, D > 0) is
 * quasi-definite whenever W > 0, so an LDL' without pivoting exists; inertia is judged on the total
 * pivot signs (n positive, m negative) and a wrong count raises delta_w.  A fixed primal-dual
 * regularisation (1e-8 on both diagonal blocks) keeps the pivots of rank-deficient row sets away
 * from round-off.  Steps are plain fraction-to-boundary lengths (an Armijo search on the barrier
 * merit was tried and rejected: it failed more sub-problems than it rescued).
 * Multipliers are kept in the JuMP sign: stationarity reads  H p + c = J'y + zl - zu.
 */
#include "sqp_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ROW_FREE 0
#define ROW_EQ 1
#define ROW_INEQ 2

struct ora_qp {
    int64_t n, m, nlin, nnzj, nnzh, N, ld;
    int64_t *jcolptr, *jrowval, *hcolptr, *hrowval;
    double *xL, *xU, *gL, *gU;
    ora_options opt;
    /* canonical problem */
    double *c, *hv, *hd, *jv, *lb, *ub, *lo, *hi, *wp, *wm;
    int *rtype, *hard;
    double sf;
    /* iterate */
    double *p, *zl, *zu, *s, *tp, *tm, *y, *vl, *vu;
    double *zp, *zm;   /* elastic duals wp - y, wm + y, carried explicitly (no cancellation near 0) */
    /* directions: affine and final */
    double *dp, *dzl, *dzu, *ds, *dtp, *dtm, *dy, *dvl, *dvu;
    /* second-order correction products */
    double *k_gl, *k_gu, *k_al, *k_au, *k_tp, *k_tm;
    /* linear algebra */
    double *K, *dinv, *rhs, *sol, *res, *sigp, *D;
    /* condensed form (opt.kkt_condense): rows with gL == gU are kept (kpos = position among them, else -1),
     * all other rows are eliminated; CSR view of J for the row-wise products */
    int64_t mk, Nc;
    int64_t *kpos, *jrowptr, *jrcol, *jrslot;
    int64_t *ord;      /* rank of unknown u (variable j, or n + kpos) in the factorised matrix (ora_set_kkt_order) */
    double *rhsc;
    /* sparse linear algebra (opt.kkt_mode): the Newton matrix as triplets in a fixed enumeration order (kkt_triplets) */
    int sparse;
    ora_sldl *sl;
    int64_t nt;
    double *tv;
    double delta_w_last;
    int cur_mode;           /* mode of the sub-problem being solved (ora_qp_solve) */
    /* warm start (opt.ipm_warm_start): mode of the last solved sub-problem, -1 = none; its p and y are still in place */
    int prev_mode;
    double prev_sf;
    /* stats */
    int ipm_iters, n_factor;
    double last_elastic;
    int last_rule;          /* how the last interior-point run ended: 0 error <= tol, 1 / 2 / 3 acceptable-termination rules, -1 not converged */
    double last_e0;         /* ... and its scaled optimality error there */
};

static double *dalloc(int64_t k) { return (double *)calloc((size_t)(k > 0 ? k : 1), sizeof(double)); }

/* ordering of the condensed matrix (test hook: the rank of every unknown in the order the product library uses,
 * sqphip_kkt_order with the padding squeezed out); applies to every ora_qp created afterwards whose condensed order
 * matches `len`; len = 0 restores the natural order (variables, then kept rows) */
static int64_t *g_kkt_rank = NULL;
static int64_t g_kkt_len = 0;
void ora_set_kkt_order(const int32_t *rank, int64_t len)
{
    free(g_kkt_rank); g_kkt_rank = NULL; g_kkt_len = 0;
    if (len <= 0 || !rank) return;
    g_kkt_rank = (int64_t *)malloc(sizeof(int64_t) * (size_t)len);
    for (int64_t i = 0; i < len; ++i) g_kkt_rank[i] = rank[i];
    g_kkt_len = len;
}

static void kkt_sparse_setup(ora_qp *q);

ora_qp *ora_qp_create(int64_t n, int64_t m, int64_t num_linear,
                      const int64_t *jcolptr, const int64_t *jrowval,
                      const int64_t *hcolptr, const int64_t *hrowval,
                      const double *xL, const double *xU, const double *gL, const double *gU,
                      const ora_options *opt)
{
    ora_qp *q = (ora_qp *)calloc(1, sizeof(ora_qp));
    q->n = n; q->m = m; q->nlin = num_linear;
    q->nnzj = jcolptr[n];
    q->nnzh = hcolptr ? hcolptr[n] : 0;
    q->N = n + m;
    q->ld = (q->N + 7) / 8 * 8;
    q->jcolptr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    memcpy(q->jcolptr, jcolptr, sizeof(int64_t) * (size_t)(n + 1));
    q->jrowval = (int64_t *)malloc(sizeof(int64_t) * (size_t)(q->nnzj + 1));
    memcpy(q->jrowval, jrowval, sizeof(int64_t) * (size_t)q->nnzj);
    q->hcolptr = (int64_t *)calloc((size_t)(n + 1), sizeof(int64_t));
    q->hrowval = (int64_t *)malloc(sizeof(int64_t) * (size_t)(q->nnzh + 1));
    if (q->nnzh) {
        memcpy(q->hcolptr, hcolptr, sizeof(int64_t) * (size_t)(n + 1));
        memcpy(q->hrowval, hrowval, sizeof(int64_t) * (size_t)q->nnzh);
    }
    q->xL = dalloc(n); q->xU = dalloc(n); q->gL = dalloc(m); q->gU = dalloc(m);
    memcpy(q->xL, xL, sizeof(double) * (size_t)n); memcpy(q->xU, xU, sizeof(double) * (size_t)n);
    memcpy(q->gL, gL, sizeof(double) * (size_t)m); memcpy(q->gU, gU, sizeof(double) * (size_t)m);
    q->opt = *opt;
    q->c = dalloc(n); q->hv = dalloc(q->nnzh); q->hd = dalloc(n); q->jv = dalloc(q->nnzj);
    q->lb = dalloc(n); q->ub = dalloc(n); q->lo = dalloc(m); q->hi = dalloc(m);
    q->wp = dalloc(m); q->wm = dalloc(m);
    q->rtype = (int *)calloc((size_t)(m + 1), sizeof(int));
    q->hard = (int *)calloc((size_t)(m + 1), sizeof(int));
    q->p = dalloc(n); q->zl = dalloc(n); q->zu = dalloc(n);
    q->s = dalloc(m); q->tp = dalloc(m); q->tm = dalloc(m); q->y = dalloc(m);
    q->vl = dalloc(m); q->vu = dalloc(m); q->zp = dalloc(m); q->zm = dalloc(m);
    q->dp = dalloc(n); q->dzl = dalloc(n); q->dzu = dalloc(n);
    q->ds = dalloc(m); q->dtp = dalloc(m); q->dtm = dalloc(m); q->dy = dalloc(m);
    q->dvl = dalloc(m); q->dvu = dalloc(m);
    q->k_gl = dalloc(n); q->k_gu = dalloc(n); q->k_al = dalloc(m); q->k_au = dalloc(m);
    q->k_tp = dalloc(m); q->k_tm = dalloc(m);
    q->dinv = dalloc(q->N);
    q->rhs = dalloc(q->N); q->sol = dalloc(q->N); q->res = dalloc(q->N);
    q->sigp = dalloc(n); q->D = dalloc(m);
    q->kpos = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m + 1));
    q->mk = 0;
    {   /* rows that stay in the condensed matrix: the equalities, and rows with more than 32 entries (eliminating
         * such a row would put a clique of that size into the matrix) -- the same rule as the product */
        int64_t *cnt = (int64_t *)calloc((size_t)(m + 1), sizeof(int64_t));
        for (int64_t k = 0; k < q->nnzj; ++k) cnt[jrowval[k]]++;
        for (int64_t i = 0; i < m; ++i) q->kpos[i] = (gL[i] == gU[i] || cnt[i] > 32) ? q->mk++ : -1;
        free(cnt);
    }
    q->Nc = n + q->mk;
    q->ord = (int64_t *)malloc(sizeof(int64_t) * (size_t)(q->Nc + 1));
    for (int64_t u = 0; u < q->Nc; ++u) q->ord[u] = (g_kkt_len == q->Nc) ? g_kkt_rank[u] : u;
    q->rhsc = dalloc(q->Nc);
    q->jrowptr = (int64_t *)calloc((size_t)(m + 2), sizeof(int64_t));
    q->jrcol = (int64_t *)malloc(sizeof(int64_t) * (size_t)(q->nnzj + 1));
    q->jrslot = (int64_t *)malloc(sizeof(int64_t) * (size_t)(q->nnzj + 1));
    for (int64_t k = 0; k < q->nnzj; ++k) q->jrowptr[jrowval[k] + 1]++;
    for (int64_t i = 0; i < m; ++i) q->jrowptr[i + 1] += q->jrowptr[i];
    {
        int64_t *fill = (int64_t *)calloc((size_t)(m + 1), sizeof(int64_t));
        for (int64_t j = 0; j < n; ++j)
            for (int64_t k = jcolptr[j]; k < jcolptr[j + 1]; ++k) {
                int64_t i = jrowval[k], t = q->jrowptr[i] + fill[i]++;
                q->jrcol[t] = j; q->jrslot[t] = k;
            }
        free(fill);
    }
    {
        const int64_t Nf = q->opt.kkt_condense ? q->Nc : q->N;
        q->sparse = q->opt.kkt_mode == 2 || (q->opt.kkt_mode == 0 && Nf >= 3000);
    }
    if (q->sparse) kkt_sparse_setup(q);
    else q->K = dalloc(q->ld * q->N);
    q->prev_mode = -1;
    return q;
}

void ora_qp_destroy(ora_qp *q)
{
    if (!q) return;
    void *ptrs[] = { q->jcolptr, q->jrowval, q->hcolptr, q->hrowval, q->xL, q->xU, q->gL, q->gU,
        q->c, q->hv, q->hd, q->jv, q->lb, q->ub, q->lo, q->hi, q->wp, q->wm, q->rtype, q->hard,
        q->p, q->zl, q->zu, q->s, q->tp, q->tm, q->y, q->vl, q->vu, q->dp, q->dzl, q->dzu, q->ds,
        q->dtp, q->dtm, q->dy, q->dvl, q->dvu, q->k_gl, q->k_gu, q->k_al, q->k_au, q->k_tp, q->k_tm,
        q->K, q->dinv, q->rhs, q->sol, q->res, q->sigp, q->D, q->zp, q->zm,
        q->kpos, q->jrowptr, q->jrcol, q->jrslot, q->rhsc, q->ord, q->tv };
    for (size_t i = 0; i < sizeof(ptrs) / sizeof(ptrs[0]); ++i) free(ptrs[i]);
    ora_sldl_free(q->sl);
    free(q);
}

/* how the last interior-point run of the last sub-problem ended (the device twin: sqphip_qp_termination) */
void ora_qp_termination(const ora_qp *q, int *rule, double *scaled_error)
{
    if (rule) *rule = q->last_rule;
    if (scaled_error) *scaled_error = q->last_e0;
}

void ora_qp_stats(const ora_qp *q, int *ipm_iters, int *n_factor, double *last_elastic)
{
    if (ipm_iters) *ipm_iters = q->ipm_iters;
    if (n_factor) *n_factor = q->n_factor;
    if (last_elastic) *last_elastic = q->last_elastic;
}

/* ------------------------------------------------------------------ dense LDL' (no pivoting) */
#define LDLT_NB 64

void ora_ldlt_factor(int64_t N, double *A, int64_t ld, double *dinv, int64_t n1,
                     int64_t *npos1, int64_t *nneg2, int nthreads)
{
    double *Wp = (double *)malloc(sizeof(double) * (size_t)(N * LDLT_NB + 8));
    double dl[LDLT_NB];
    (void)nthreads;
    for (int64_t k0 = 0; k0 < N; k0 += LDLT_NB) {
        int64_t kb = N - k0 < LDLT_NB ? N - k0 : LDLT_NB;
        /* diagonal block, left-looking by column */
        for (int64_t j = 0; j < kb; ++j) {
            double *cj = A + (k0 + j) * ld;
            for (int64_t cc = 0; cc < j; ++cc) {
                const double *lc = A + (k0 + cc) * ld;
                double f = lc[k0 + j] / dinv[k0 + cc];   /* d_c * L_jc */
                for (int64_t i = k0 + j; i < k0 + kb; ++i) cj[i] -= lc[i] * f;
            }
            double d = cj[k0 + j];
            dinv[k0 + j] = 1.0 / d;
            for (int64_t i = k0 + j + 1; i < k0 + kb; ++i) cj[i] *= dinv[k0 + j];
        }
        int64_t r0 = k0 + kb;
        if (r0 >= N) break;
        /* rows below: X <- X * (L_kk D)^-T, independent per row chunk */
        int64_t nrows = N - r0;
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
        for (int64_t c0 = 0; c0 < nrows; c0 += 256) {
            int64_t c1 = c0 + 256 < nrows ? c0 + 256 : nrows;
            for (int64_t j = 0; j < kb; ++j) {
                double *cj = A + (k0 + j) * ld + r0;
                for (int64_t cc = 0; cc < j; ++cc) {
                    const double *lc = A + (k0 + cc) * ld + r0;
                    double f = A[(k0 + cc) * ld + k0 + j] / dinv[k0 + cc];
                    for (int64_t i = c0; i < c1; ++i) cj[i] -= lc[i] * f;
                }
                double di = dinv[k0 + j];
                for (int64_t i = c0; i < c1; ++i) cj[i] *= di;
            }
        }
        /* W = L_panel * D */
        for (int64_t j = 0; j < kb; ++j) {
            dl[j] = 1.0 / dinv[k0 + j];
            const double *lj = A + (k0 + j) * ld + r0;
            double *wj = Wp + j * nrows;
            for (int64_t i = 0; i < nrows; ++i) wj[i] = lj[i] * dl[j];
        }
        /* trailing update A22 -= W L21', four columns at a time (rows from the first column's
         * diagonal: the few entries written above the diagonal are scratch, never read) */
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
        for (int64_t jb = 0; jb < nrows; jb += 4) {
            int64_t jw = nrows - jb < 4 ? nrows - jb : 4;
            double *a0 = A + (r0 + jb) * ld + r0;
            double *a1 = a0 + ld, *a2 = a0 + 2 * ld, *a3 = a0 + 3 * ld;
            if (jw == 4) {
                for (int64_t cc = 0; cc < kb; ++cc) {
                    const double *w = Wp + cc * nrows;
                    const double *lrow = A + (k0 + cc) * ld + r0 + jb;
                    double l0 = lrow[0], l1 = lrow[1], l2 = lrow[2], l3 = lrow[3];
                    for (int64_t i = jb; i < nrows; ++i) {
                        double wi = w[i];
                        a0[i] -= wi * l0; a1[i] -= wi * l1; a2[i] -= wi * l2; a3[i] -= wi * l3;
                    }
                }
            } else {
                for (int64_t jj = 0; jj < jw; ++jj) {
                    double *aj = a0 + jj * ld;
                    for (int64_t cc = 0; cc < kb; ++cc) {
                        const double *w = Wp + cc * nrows;
                        double l = A[(k0 + cc) * ld + r0 + jb + jj];
                        for (int64_t i = jb + jj; i < nrows; ++i) aj[i] -= w[i] * l;
                    }
                }
            }
        }
    }
    free(Wp);
    int64_t np = 0, nn = 0;
    for (int64_t j = 0; j < N; ++j) {
        double d = dinv[j];
        if (j < n1) { if (isfinite(d) && d > 0) ++np; }
        else { if (isfinite(d) && d < 0) ++nn; }
    }
    if (npos1) *npos1 = np;
    if (nneg2) *nneg2 = nn;
}

void ora_ldlt_solve(int64_t N, const double *A, int64_t ld, const double *dinv, double *x)
{
    for (int64_t j = 0; j < N; ++j) {
        double xj = x[j];
        const double *lj = A + j * ld;
        for (int64_t i = j + 1; i < N; ++i) x[i] -= lj[i] * xj;
    }
    for (int64_t j = 0; j < N; ++j) x[j] *= dinv[j];
    for (int64_t j = N - 1; j >= 0; --j) {
        const double *lj = A + j * ld;
        double acc = 0.0;
        for (int64_t i = j + 1; i < N; ++i) acc += lj[i] * x[i];
        x[j] -= acc;
    }
}

/* ------------------------------------------------------------------ sparse helpers */
/* out = (H + diag(hd) + diag(extra)) v   with H full symmetric CSC */
static void hess_mul(const ora_qp *q, const double *extra, const double *v, double *out)
{
    for (int64_t j = 0; j < q->n; ++j) out[j] = (q->hd[j] + (extra ? extra[j] : 0.0)) * v[j];
    for (int64_t j = 0; j < q->n; ++j) {
        double vj = v[j];
        if (vj == 0.0) continue;
        for (int64_t k = q->hcolptr[j]; k < q->hcolptr[j + 1]; ++k) out[q->hrowval[k]] += q->hv[k] * vj;
    }
}
/* out_i = J_i v for active rows */
static void jac_mul(const ora_qp *q, const double *v, double *out)
{
    memset(out, 0, sizeof(double) * (size_t)q->m);
    for (int64_t j = 0; j < q->n; ++j) {
        double vj = v[j];
        if (vj == 0.0) continue;
        for (int64_t k = q->jcolptr[j]; k < q->jcolptr[j + 1]; ++k) out[q->jrowval[k]] += q->jv[k] * vj;
    }
    for (int64_t i = 0; i < q->m; ++i) if (q->rtype[i] == ROW_FREE) out[i] = 0.0;
}
/* out_j += sign * J' w  over active rows */
static void jact_mul_add(const ora_qp *q, const double *w, double sign, double *out)
{
    for (int64_t j = 0; j < q->n; ++j) {
        double acc = 0.0;
        for (int64_t k = q->jcolptr[j]; k < q->jcolptr[j + 1]; ++k) {
            int64_t i = q->jrowval[k];
            if (q->rtype[i] != ROW_FREE) acc += q->jv[k] * w[i];
        }
        out[j] += sign * acc;
    }
}

/* ------------------------------------------------------------------ KKT assembly / solve */
/* primal-dual regularisation of the Newton system (part of the method, not of the residuals) */
static double ipm_reg_p(void) { return 1e-8; }
static double ipm_reg_d(void) { return 1e-8; }

static void kkt_assemble(ora_qp *q, double delta_w)
{
    int64_t n = q->n, m = q->m, N = q->N, ld = q->ld;
    memset(q->K, 0, sizeof(double) * (size_t)(ld * N));
    for (int64_t j = 0; j < n; ++j) {
        double *col = q->K + j * ld;
        col[j] = q->hd[j] + q->sigp[j] + delta_w + ipm_reg_p();
        for (int64_t k = q->hcolptr[j]; k < q->hcolptr[j + 1]; ++k) {
            int64_t i = q->hrowval[k];
            if (i >= j) col[i] += q->hv[k];
        }
        for (int64_t k = q->jcolptr[j]; k < q->jcolptr[j + 1]; ++k) {
            int64_t i = q->jrowval[k];
            if (q->rtype[i] != ROW_FREE) col[n + i] += q->jv[k];
        }
    }
    for (int64_t i = 0; i < m; ++i)
        q->K[(n + i) * ld + n + i] = q->rtype[i] == ROW_FREE ? -1.0 : -(q->D[i] + ipm_reg_d());
}

/* Condensed Newton matrix: the rows with gL != gU have the diagonal block -(D + reg) and are eliminated exactly,
 *   [ W + J_I' (D_I + reg)^-1 J_I    J_E' ]   order n + mk,
 *   [ J_E                           -(D_E + reg) ]
 * same inertia rule (n positive pivots) by Haynsworth's additivity, since the eliminated block is negative definite. */
static void kkt_assemble_condensed(ora_qp *q, double delta_w)
{
    int64_t n = q->n, m = q->m, Nc = q->Nc, ld = q->ld;
    const int64_t *ord = q->ord;
    memset(q->K, 0, sizeof(double) * (size_t)(ld * Nc));
    /* entry (a, b) of the symmetric matrix, in the order `ord`, lower triangle */
#define KADD(a, b, v) do { int64_t ra_ = ord[a], rb_ = ord[b]; \
        if (ra_ >= rb_) q->K[rb_ * ld + ra_] += (v); else q->K[ra_ * ld + rb_] += (v); } while (0)
    for (int64_t j = 0; j < n; ++j) {
        KADD(j, j, q->hd[j] + q->sigp[j] + delta_w + ipm_reg_p());
        for (int64_t k = q->hcolptr[j]; k < q->hcolptr[j + 1]; ++k) {
            int64_t i = q->hrowval[k];
            if (i >= j) KADD(i, j, q->hv[k]);
        }
        for (int64_t k = q->jcolptr[j]; k < q->jcolptr[j + 1]; ++k) {
            int64_t i = q->jrowval[k];
            if (q->rtype[i] == ROW_FREE) continue;
            if (q->kpos[i] >= 0) { KADD(n + q->kpos[i], j, q->jv[k]); continue; }
            const double f = q->jv[k] / (q->D[i] + ipm_reg_d());
            for (int64_t t = q->jrowptr[i]; t < q->jrowptr[i + 1]; ++t)
                if (q->jrcol[t] >= j) KADD(q->jrcol[t], j, f * q->jv[q->jrslot[t]]);
        }
    }
    for (int64_t i = 0; i < m; ++i)
        if (q->kpos[i] >= 0)
            KADD(n + q->kpos[i], n + q->kpos[i], q->rtype[i] == ROW_FREE ? -1.0 : -(q->D[i] + ipm_reg_d()));
#undef KADD
}

/* The same matrix (full or condensed form) as triplets over the unknowns u = variable j, or n + (row | kept position),
 * in a fixed enumeration order; free rows keep their structural entries with value 0 (diagonal -1) so that the
 * pattern does not depend on the mode.  ti / tj / tv: whichever is non-NULL is filled; returns the count. */
static int64_t kkt_triplets(const ora_qp *q, double delta_w, int64_t *ti, int64_t *tj, double *tv)
{
    const int64_t n = q->n, m = q->m;
    const int cond = q->opt.kkt_condense;
    int64_t t = 0;
#define TADD(a, b, v) do { if (ti) { ti[t] = (a); tj[t] = (b); } if (tv) tv[t] = (v); ++t; } while (0)
    for (int64_t j = 0; j < n; ++j) {
        TADD(j, j, tv ? q->hd[j] + q->sigp[j] + delta_w + ipm_reg_p() : 0.0);
        for (int64_t k = q->hcolptr[j]; k < q->hcolptr[j + 1]; ++k) {
            const int64_t i = q->hrowval[k];
            if (i >= j) TADD(i, j, tv ? q->hv[k] : 0.0);
        }
    }
    for (int64_t i = 0; i < m; ++i) {
        const int act = tv ? q->rtype[i] != ROW_FREE : 1;
        const int64_t u = cond ? (q->kpos[i] >= 0 ? n + q->kpos[i] : -1) : n + i;
        if (u >= 0) {
            TADD(u, u, tv ? (act ? -(q->D[i] + ipm_reg_d()) : -1.0) : 0.0);
            for (int64_t s = q->jrowptr[i]; s < q->jrowptr[i + 1]; ++s) TADD(u, q->jrcol[s], (tv && act) ? q->jv[q->jrslot[s]] : 0.0);
        } else {
            const double f = (tv && act) ? 1.0 / (q->D[i] + ipm_reg_d()) : 0.0;
            for (int64_t a = q->jrowptr[i]; a < q->jrowptr[i + 1]; ++a)
                for (int64_t b = q->jrowptr[i]; b <= a; ++b)
                    TADD(q->jrcol[a], q->jrcol[b], (tv && act) ? q->jv[q->jrslot[a]] * f * q->jv[q->jrslot[b]] : 0.0);
        }
    }
#undef TADD
    return t;
}

static void kkt_sparse_setup(ora_qp *q)
{
    const int64_t n = q->n, m = q->m, Nf = q->opt.kkt_condense ? q->Nc : q->N;
    q->nt = kkt_triplets(q, 0.0, NULL, NULL, NULL);
    int64_t *ti = (int64_t *)malloc(sizeof(int64_t) * (size_t)(q->nt + 1)), *tj = (int64_t *)malloc(sizeof(int64_t) * (size_t)(q->nt + 1));
    kkt_triplets(q, 0.0, ti, tj, NULL);
    /* a row of the matrix is eliminated behind every variable it couples to */
    int64_t *bp = (int64_t *)calloc((size_t)(Nf + 2), sizeof(int64_t)), *bi = (int64_t *)malloc(sizeof(int64_t) * (size_t)(q->nnzj + 1));
    int64_t o = 0;
    for (int64_t u = 0; u < n; ++u) bp[u + 1] = 0;
    for (int64_t i = 0; i < m; ++i) {
        const int64_t u = q->opt.kkt_condense ? (q->kpos[i] >= 0 ? n + q->kpos[i] : -1) : n + i;
        if (u < 0) continue;
        for (int64_t s = q->jrowptr[i]; s < q->jrowptr[i + 1]; ++s) bi[o++] = q->jrcol[s];
        bp[u + 1] = o;          /* rows are visited in ascending u, variables (u < n) have empty lists */
    }
    for (int64_t u = n; u < Nf; ++u) if (bp[u + 1] < bp[u]) bp[u + 1] = bp[u];
    {
        const char *ra = getenv("ORA_ROWS_AFTER");
        const int unconstrained = ra && atoi(ra) == 0;
        q->sl = ora_sldl_analyse(Nf, q->nt, ti, tj, unconstrained ? NULL : bp, unconstrained ? NULL : bi, 0);
        if (getenv("ORA_SYM_STATS")) fprintf(stderr, "oracle: sparse order %ld nnz(L) %ld\n", (long)Nf, (long)ora_sldl_nnz_l(q->sl));
    }
    q->tv = dalloc(q->nt);
    free(ti); free(tj); free(bp); free(bi);
}

/* res = rhs - K sol  with K applied through its sparse pieces */
static double kkt_residual(const ora_qp *q, double delta_w, const double *rhs, const double *sol,
                           double *res)
{
    int64_t n = q->n, m = q->m;
    double *tmp = (double *)malloc(sizeof(double) * (size_t)(q->N));
    double *ext = (double *)malloc(sizeof(double) * (size_t)n);
    for (int64_t j = 0; j < n; ++j) ext[j] = q->sigp[j] + delta_w + ipm_reg_p();
    hess_mul(q, ext, sol, tmp);
    jact_mul_add(q, sol + n, 1.0, tmp);
    jac_mul(q, sol, tmp + n);
    double nrm = 0.0;
    for (int64_t j = 0; j < n; ++j) { res[j] = rhs[j] - tmp[j]; nrm = fmax(nrm, fabs(res[j])); }
    for (int64_t i = 0; i < m; ++i) {
        double d = q->rtype[i] == ROW_FREE ? 1.0 : q->D[i] + ipm_reg_d();
        res[n + i] = rhs[n + i] - (tmp[n + i] - d * sol[n + i]);
        nrm = fmax(nrm, fabs(res[n + i]));
    }
    free(tmp); free(ext);
    return nrm;
}

/* factor with inertia correction (Ipopt-style delta_w schedule, starting from dw_floor);
 * inertia is judged on the TOTAL pivot signs (Sylvester): n positive, m negative.
 * returns 0 on success */
static int kkt_factor(ora_qp *q, double dw_floor, double *delta_w_out)
{
    double dw = dw_floor;
    for (int attempt = 0; attempt < 60; ++attempt) {
        const int64_t Nf = q->opt.kkt_condense ? q->Nc : q->N;
        int64_t np = 0, nn = 0, bad = 0;
        if (q->sparse) {
            kkt_triplets(q, dw, NULL, NULL, q->tv);
            np = ora_sldl_numeric(q->sl, q->tv, &bad);
        } else {
            if (q->opt.kkt_condense) kkt_assemble_condensed(q, dw); else kkt_assemble(q, dw);
            ora_ldlt_factor(Nf, q->K, q->ld, q->dinv, Nf, &np, &nn, q->opt.num_threads);
        }
        q->n_factor++;
        if (np == q->n) {
            if (!q->sparse) for (int64_t j = 0; j < Nf; ++j) if (!isfinite(q->dinv[j]) || q->dinv[j] == 0.0) ++bad;
            if (!bad) {
                if (q->sparse && getenv("ORA_KKT_DUMP")) {      /* experiment aid: every 16th accepted matrix as triplets */
                    static long ndump = 0;
                    if (ndump++ % 16 == 0) {
                        char path[512];
                        snprintf(path, sizeof path, "%s_%05ld.bin", getenv("ORA_KKT_DUMP"), ndump / 16);
                        FILE *fh = fopen(path, "wb");
                        if (fh) {
                            int64_t *ti = (int64_t *)malloc(sizeof(int64_t) * (size_t)q->nt), *tj = (int64_t *)malloc(sizeof(int64_t) * (size_t)q->nt);
                            kkt_triplets(q, dw, ti, tj, NULL);
                            int64_t hdr[4] = { Nf, q->nt, q->n, (int64_t)attempt };
                            fwrite(hdr, sizeof(int64_t), 4, fh); fwrite(ti, sizeof(int64_t), (size_t)q->nt, fh);
                            fwrite(tj, sizeof(int64_t), (size_t)q->nt, fh); fwrite(q->tv, sizeof(double), (size_t)q->nt, fh);
                            fclose(fh); free(ti); free(tj);
                        }
                    }
                }
                if (q->sparse && getenv("ORA_INERTIA_STATS")) {  /* experiment aid: signs of the VARIABLE pivots of accepted matrices */
                    static long nacc = 0, nacc_neg = 0, nneg = 0, nacc_dw = 0, nacc_dw_neg = 0;
                    const int64_t *perm = ora_sldl_perm(q->sl);
                    const double *Dp = ora_sldl_pivots(q->sl);
                    long neg = 0;
                    for (int64_t k = 0; k < Nf; ++k) if (perm[k] < q->n && Dp[k] < 0.0) ++neg;
                    ++nacc; nneg += neg; if (neg) ++nacc_neg;
                    if (dw > 0.0) { ++nacc_dw; if (neg) ++nacc_dw_neg; }
                    if (nacc % 500 == 0)
                        fprintf(stderr, "[inertia-stats] accepted %ld, with negative variable pivots %ld (sum %ld); with dw > 0: %ld, of those with negative variable pivots %ld\n",
                                nacc, nacc_neg, nneg, nacc_dw, nacc_dw_neg);
                }
                if (dw > 0.0) q->delta_w_last = dw;
                *delta_w_out = dw;
                return 0;
            }
        }
        if (dw == 0.0) dw = q->delta_w_last == 0.0 ? 1e-4 : fmax(1e-20, q->delta_w_last / 3.0);
        else {
            double grow = q->delta_w_last == 0.0 ? 100.0 : 8.0;
            if (getenv("ORA_DW_ADAPT")) {       /* experiment: after k failed trials of one call grow by g instead */
                int k = 2; double g = 64.0;
                sscanf(getenv("ORA_DW_ADAPT"), "%d,%lf", &k, &g);
                if (attempt >= k) grow = g;
            }
            dw *= grow;
        }
        if (dw > 1e40) break;
    }
    if (getenv("ORA_IPM_DEBUG")) {
        int64_t p1 = 0, p2 = 0, bad = 0;
        for (int64_t j = 0; j < q->N; ++j) {
            if (!isfinite(q->dinv[j]) || q->dinv[j] == 0.0) { if (bad < 4) fprintf(stderr, "   bad pivot %ld dinv=%g\n", (long)j, q->dinv[j]); ++bad; }
            else if (q->dinv[j] > 0) { if (j < q->n) ++p1; else ++p2; }
        }
        fprintf(stderr, "   kkt_factor gave up: dw=%.1e pos(W)=%ld/%ld pos(rows)=%ld bad=%ld\n", dw, (long)p1, (long)q->n, (long)p2, (long)bad);
    }
    return -1;
}

/* experiment counters (not thread safe: run single-threaded when reading them) */
static double g_dbg_solves, g_dbg_refines, g_dbg_maxres0, g_dbg_maxres1, g_dbg_bad0, g_dbg_bad1;
void ora_dbg_counts(double *out) { out[0] = g_dbg_solves; out[1] = g_dbg_refines; out[2] = g_dbg_maxres0; out[3] = g_dbg_maxres1; out[4] = g_dbg_bad0; out[5] = g_dbg_bad1;
    g_dbg_solves = g_dbg_refines = g_dbg_maxres0 = g_dbg_maxres1 = g_dbg_bad0 = g_dbg_bad1 = 0.0; }

/* sol = K^-1 rhs through the factors in q->K (full or condensed form) */
static void kkt_apply(ora_qp *q, const double *rhs, double *sol)
{
    int64_t N = q->N;
    if (q->opt.kkt_condense) {
        /* condensed right-hand side g + J_I' (D_I + reg)^-1 b_I, solve, then q_I = (J_I dp - b_I) / (D_I + reg) */
        int64_t n = q->n, m = q->m;
        double *rc = q->rhsc;
        const int64_t *ord = q->ord;
        for (int64_t j = 0; j < n; ++j) {
            double acc = 0.0;
            for (int64_t k = q->jcolptr[j]; k < q->jcolptr[j + 1]; ++k) {
                int64_t i = q->jrowval[k];
                if (q->rtype[i] != ROW_FREE && q->kpos[i] < 0) acc += q->jv[k] * rhs[n + i] / (q->D[i] + ipm_reg_d());
            }
            rc[q->sparse ? j : ord[j]] = rhs[j] + acc;
        }
        for (int64_t i = 0; i < m; ++i) if (q->kpos[i] >= 0) rc[q->sparse ? n + q->kpos[i] : ord[n + q->kpos[i]]] = rhs[n + i];
        if (q->sparse) ora_sldl_solve(q->sl, rc);
        else ora_ldlt_solve(q->Nc, q->K, q->ld, q->dinv, rc);
        for (int64_t j = 0; j < n; ++j) sol[j] = rc[q->sparse ? j : ord[j]];
        for (int64_t i = 0; i < m; ++i) {
            if (q->kpos[i] >= 0) { sol[n + i] = rc[q->sparse ? n + q->kpos[i] : ord[n + q->kpos[i]]]; continue; }
            if (q->rtype[i] == ROW_FREE) { sol[n + i] = -rhs[n + i]; continue; }     /* row -1 * q = b */
            double acc = 0.0;
            for (int64_t t = q->jrowptr[i]; t < q->jrowptr[i + 1]; ++t) acc += q->jv[q->jrslot[t]] * sol[q->jrcol[t]];
            sol[n + i] = (acc - rhs[n + i]) / (q->D[i] + ipm_reg_d());
        }
    } else {
        memcpy(sol, rhs, sizeof(double) * (size_t)N);
        if (q->sparse) ora_sldl_solve(q->sl, sol);
        else ora_ldlt_solve(N, q->K, q->ld, q->dinv, sol);
    }
}

/* solve, residual against the sparse operator; returns |res|/max(1,|rhs|) */
static double kkt_solve(ora_qp *q, double delta_w, const double *rhs, double *sol, int no_refine)
{
    int64_t N = q->N;
    kkt_apply(q, rhs, sol);
    double rn = 0.0, en = 0.0;
    for (int64_t i = 0; i < N; ++i) rn = fmax(rn, fabs(rhs[i]));
    rn = fmax(1.0, rn);
    /* Full form: no iterative refinement.  With the 1e-8 regularisation inside the factorised matrix the plain solve is
     * accurate to ~1e-12 relative on 94 % of the systems and to 1e-8 on the rest, and one refinement step (the
     * policy until late in round 1) changed no iteration count on any test problem (4224 IPM iterations over the
     * IEEE-14 contingency set, 683 on IEEE-118, with or without it).  The residual is still measured against the
     * sparse operator: a direction above 1e-6 relative is rejected by the caller (delta_w escalation).
     * Condensed form: the elimination puts 1/D-sized terms (up to 1e8) into the matrix and the plain solve loses the
     * digits two implementations need to stay on one trajectory; one step of refinement against the FULL sparse
     * operator, taken when the residual is above 1e-11 relative, restores them. */
    en = kkt_residual(q, delta_w, rhs, sol, q->res);
    g_dbg_solves += 1.0; if (en / rn > g_dbg_maxres0) g_dbg_maxres0 = en / rn;
    if (en / rn > 1e-8) g_dbg_bad0 += 1.0;
    const char *rt_env = getenv("ORA_REFINE_TOL");
    const double rtol = rt_env ? atof(rt_env) : 1e-11;
    if (q->opt.kkt_condense && !no_refine && en > rtol * rn) {
        double *corr = (double *)malloc(sizeof(double) * (size_t)N);
        double *r0 = (double *)malloc(sizeof(double) * (size_t)N);
        memcpy(r0, q->res, sizeof(double) * (size_t)N);
        kkt_apply(q, r0, corr);
        for (int64_t i = 0; i < N; ++i) sol[i] += corr[i];
        en = kkt_residual(q, delta_w, rhs, sol, q->res);
        free(corr); free(r0);
        g_dbg_refines += 1.0;
    }
    if (en / rn > g_dbg_maxres1) g_dbg_maxres1 = en / rn;
    if (en / rn > 1e-10) g_dbg_bad1 += 1.0;
    return en / rn;
}

/* ------------------------------------------------------------------ interior point */
static double push_inside(double v, double lo, double hi)
{
    const double k1 = 1e-2, k2 = 1e-2;
    int hl = isfinite(lo), hu = isfinite(hi);
    if (hl && hu) {
        double w = hi - lo;
        double pl = fmin(k1 * fmax(1.0, fabs(lo)), k2 * w);
        double pu = fmin(k1 * fmax(1.0, fabs(hi)), k2 * w);
        if (v < lo + pl) v = lo + pl;
        if (v > hi - pu) v = hi - pu;
    } else if (hl) {
        double pl = k1 * fmax(1.0, fabs(lo));
        if (v < lo + pl) v = lo + pl;
    } else if (hu) {
        double pu = k1 * fmax(1.0, fabs(hi));
        if (v > hi - pu) v = hi - pu;
    }
    return v;
}

/* residuals and error measures at the current iterate */
typedef struct { double rd, rp, cavg, cmax, dual_l1; int64_t ncomp; } ipm_meas;

static void ipm_measure(ora_qp *q, double *rd_vec, double *rp_vec, ipm_meas *ms)
{
    int64_t n = q->n, m = q->m;
    hess_mul(q, NULL, q->p, rd_vec);
    for (int64_t j = 0; j < n; ++j) {
        rd_vec[j] += q->c[j];
        if (isfinite(q->lb[j])) rd_vec[j] -= q->zl[j];
        if (isfinite(q->ub[j])) rd_vec[j] += q->zu[j];
    }
    jact_mul_add(q, q->y, -1.0, rd_vec);
    jac_mul(q, q->p, rp_vec);
    double csum = 0.0, cmax = 0.0, rd = 0.0, rp = 0.0, dl1 = 0.0;
    int64_t nc = 0;
#define CP(zz, xx) do { double c_ = (zz) * (xx); csum += c_; cmax = fmax(cmax, c_); ++nc; } while (0)
    for (int64_t j = 0; j < n; ++j) {
        rd = fmax(rd, fabs(rd_vec[j]));
        if (isfinite(q->lb[j])) { CP(q->zl[j], q->p[j] - q->lb[j]); dl1 += q->zl[j]; }
        if (isfinite(q->ub[j])) { CP(q->zu[j], q->ub[j] - q->p[j]); dl1 += q->zu[j]; }
    }
    for (int64_t i = 0; i < m; ++i) {
        if (q->rtype[i] == ROW_FREE) { rp_vec[i] = 0.0; continue; }
        rp_vec[i] += q->tp[i] - q->tm[i] - q->s[i];
        rp = fmax(rp, fabs(rp_vec[i]));
        CP(q->zp[i], q->tp[i]);
        CP(q->zm[i], q->tm[i]);
        dl1 += fabs(q->y[i]);
        if (q->rtype[i] == ROW_INEQ) {
            if (isfinite(q->lo[i])) CP(q->vl[i], q->s[i] - q->lo[i]);
            if (isfinite(q->hi[i])) CP(q->vu[i], q->hi[i] - q->s[i]);
        }
    }
#undef CP
    ms->cavg = nc ? csum / (double)nc : 0.0;
    ms->rd = rd; ms->rp = rp; ms->cmax = cmax; ms->ncomp = nc; ms->dual_l1 = dl1;
}

/* max_k |z_k x_k - mu| over all complementarity pairs */
static double ipm_compl_err(const ora_qp *q, double mu)
{
    double e = 0.0;
#define CE(zz, xx) e = fmax(e, fabs((zz) * (xx) - mu))
    for (int64_t j = 0; j < q->n; ++j) {
        if (isfinite(q->lb[j])) CE(q->zl[j], q->p[j] - q->lb[j]);
        if (isfinite(q->ub[j])) CE(q->zu[j], q->ub[j] - q->p[j]);
    }
    for (int64_t i = 0; i < q->m; ++i) {
        if (q->rtype[i] == ROW_FREE) continue;
        CE(q->zp[i], q->tp[i]);
        CE(q->zm[i], q->tm[i]);
        if (q->rtype[i] == ROW_INEQ) {
            if (isfinite(q->lo[i])) CE(q->vl[i], q->s[i] - q->lo[i]);
            if (isfinite(q->hi[i])) CE(q->vu[i], q->hi[i] - q->s[i]);
        }
    }
#undef CE
    return e;
}

/* Build the rhs of the reduced Newton system for centring target `tgt`, solve, and expand to
 * all directions.  rd_vec / rp_vec are the current residuals.  Returns the relative residual
 * of the linear solve after refinement. */
/* soc (may be NULL): per-pair second-order terms dz_aff * dx_aff of the predictor, laid out
 * [zl n | zu n | zp m | zm m | vl m | vu m]; the complementarity targets become tgt - soc */
/* predictor != 0: the affine-scaling direction of the predictor-corrector mode -- it only feeds the centring parameter
 * and the second-order terms, so it is not refined (the corrector, the direction actually taken, is) */
static double ipm_direction(ora_qp *q, double delta_w, double tgt,
                            const double *rd_vec, const double *rp_vec, const double *soc, int predictor)
{
    int64_t n = q->n, m = q->m;
    const int64_t oZL = 0, oZU = n, oZP = 2 * n, oZM = 2 * n + m, oVL = 2 * n + 2 * m, oVU = 2 * n + 3 * m;
#define SOC(k, idx) (soc ? soc[(k) + (idx)] : 0.0)
    for (int64_t j = 0; j < n; ++j) {
        double g = -rd_vec[j];
        if (isfinite(q->lb[j])) { double gl = q->p[j] - q->lb[j]; g += (tgt - SOC(oZL, j) - q->zl[j] * gl) / gl; }
        if (isfinite(q->ub[j])) { double gu = q->ub[j] - q->p[j]; g -= (tgt - SOC(oZU, j) - q->zu[j] * gu) / gu; }
        q->rhs[j] = g;
    }
    for (int64_t i = 0; i < m; ++i) {
        if (q->rtype[i] == ROW_FREE) { q->rhs[n + i] = 0.0; continue; }
        double zp = q->zp[i], zm = q->zm[i];
        double cp = tgt - SOC(oZP, i) - zp * q->tp[i], cm = tgt - SOC(oZM, i) - zm * q->tm[i];
        double b = -rp_vec[i] - cp / zp + cm / zm;
        if (q->rtype[i] == ROW_INEQ) {
            double sig = 0.0, t = 0.0;
            if (isfinite(q->lo[i])) { double al = q->s[i] - q->lo[i]; sig += q->vl[i] / al; t += (tgt - SOC(oVL, i) - q->vl[i] * al) / al; }
            if (isfinite(q->hi[i])) { double au = q->hi[i] - q->s[i]; sig += q->vu[i] / au; t -= (tgt - SOC(oVU, i) - q->vu[i] * au) / au; }
            b += t / sig;
        }
        q->rhs[n + i] = b;
    }
    double relres = kkt_solve(q, delta_w, q->rhs, q->sol, predictor);
    for (int64_t j = 0; j < n; ++j) {
        double dp = q->sol[j];
        q->dp[j] = dp;
        q->dzl[j] = 0.0; q->dzu[j] = 0.0;
        if (isfinite(q->lb[j])) { double gl = q->p[j] - q->lb[j]; q->dzl[j] = (tgt - SOC(oZL, j) - q->zl[j] * gl - q->zl[j] * dp) / gl; }
        if (isfinite(q->ub[j])) { double gu = q->ub[j] - q->p[j]; q->dzu[j] = (tgt - SOC(oZU, j) - q->zu[j] * gu + q->zu[j] * dp) / gu; }
    }
    for (int64_t i = 0; i < m; ++i) {
        q->dy[i] = q->ds[i] = q->dtp[i] = q->dtm[i] = q->dvl[i] = q->dvu[i] = 0.0;
        if (q->rtype[i] == ROW_FREE) continue;
        double dy = -q->sol[n + i];
        double zp = q->zp[i], zm = q->zm[i];
        q->dy[i] = dy;
        q->dtp[i] = (tgt - SOC(oZP, i) - zp * q->tp[i] + q->tp[i] * dy) / zp;
        q->dtm[i] = (tgt - SOC(oZM, i) - zm * q->tm[i] - q->tm[i] * dy) / zm;
        if (q->rtype[i] == ROW_INEQ) {
            double sig = 0.0, t = 0.0, al = 0.0, au = 0.0, cl = 0.0, cu = 0.0;
            int hl = isfinite(q->lo[i]), hu = isfinite(q->hi[i]);
            if (hl) { al = q->s[i] - q->lo[i]; cl = tgt - SOC(oVL, i) - q->vl[i] * al; sig += q->vl[i] / al; t += cl / al; }
            if (hu) { au = q->hi[i] - q->s[i]; cu = tgt - SOC(oVU, i) - q->vu[i] * au; sig += q->vu[i] / au; t -= cu / au; }
            double ds = (t - dy) / sig;
            q->ds[i] = ds;
            if (hl) q->dvl[i] = (cl - q->vl[i] * ds) / al;
            if (hu) q->dvu[i] = (cu + q->vu[i] * ds) / au;
        }
    }
    return relres;
}
#undef SOC

static inline double ratio(double x, double dx, double a)
{
    if (dx < 0.0) { double r = -x / dx; if (r < a) a = r; }
    return a;
}

/* largest steps keeping primal gaps / duals positive (before the fraction-to-boundary factor) */
static void ipm_max_steps(const ora_qp *q, double *ap, double *ad)
{
    double a = 1e300, d = 1e300;
    for (int64_t j = 0; j < q->n; ++j) {
        if (isfinite(q->lb[j])) { a = ratio(q->p[j] - q->lb[j], q->dp[j], a); d = ratio(q->zl[j], q->dzl[j], d); }
        if (isfinite(q->ub[j])) { a = ratio(q->ub[j] - q->p[j], -q->dp[j], a); d = ratio(q->zu[j], q->dzu[j], d); }
    }
    for (int64_t i = 0; i < q->m; ++i) {
        if (q->rtype[i] == ROW_FREE) continue;
        a = ratio(q->tp[i], q->dtp[i], a);
        a = ratio(q->tm[i], q->dtm[i], a);
        d = ratio(q->zp[i], -q->dy[i], d);
        d = ratio(q->zm[i], q->dy[i], d);
        if (q->rtype[i] == ROW_INEQ) {
            if (isfinite(q->lo[i])) { a = ratio(q->s[i] - q->lo[i], q->ds[i], a); d = ratio(q->vl[i], q->dvl[i], d); }
            if (isfinite(q->hi[i])) { a = ratio(q->hi[i] - q->s[i], -q->ds[i], a); d = ratio(q->vu[i], q->dvu[i], d); }
        }
    }
    *ap = a; *ad = d;
}

/* keep a primal variable a few ulps inside its box: p + a dp may round onto the bound although the
 * fraction-to-boundary rule holds in exact arithmetic */
static double nudge_inside(double v, double lo, double hi)
{
    if (isfinite(lo)) { double g = 1e-15 * fmax(1.0, fabs(lo)); if (v - lo < g) v = lo + g; }
    if (isfinite(hi)) { double g = 1e-15 * fmax(1.0, fabs(hi)); if (hi - v < g) v = hi - g; }
    return v;
}

/* y_start (may be NULL): multipliers of the equality rows to start from (warm start); capped like the cold ones */
static void ipm_init(ora_qp *q, const double *p_start, const double *y_start)
{
    int64_t n = q->n, m = q->m;
    const double mu0 = 1.0;
    for (int64_t j = 0; j < n; ++j) {
        q->p[j] = push_inside(p_start ? p_start[j] : 0.0, q->lb[j], q->ub[j]);
        q->zl[j] = isfinite(q->lb[j]) ? mu0 / (q->p[j] - q->lb[j]) : 0.0;
        q->zu[j] = isfinite(q->ub[j]) ? mu0 / (q->ub[j] - q->p[j]) : 0.0;
    }
    double *v = q->res; /* scratch, length >= m */
    jac_mul(q, q->p, v);
    for (int64_t i = 0; i < m; ++i) {
        q->s[i] = q->tp[i] = q->tm[i] = q->y[i] = q->vl[i] = q->vu[i] = 0.0;
        q->zp[i] = q->zm[i] = 1.0;
        if (q->rtype[i] == ROW_FREE) continue;
        double s = q->rtype[i] == ROW_EQ ? q->lo[i] : push_inside(v[i], q->lo[i], q->hi[i]);
        double d = s - v[i];
        double y = 0.0;
        if (q->rtype[i] == ROW_INEQ) {
            if (isfinite(q->lo[i])) q->vl[i] = mu0 / (s - q->lo[i]);
            if (isfinite(q->hi[i])) q->vu[i] = mu0 / (q->hi[i] - s);
            y = q->vl[i] - q->vu[i];
            double cap = 0.5 * fmin(q->wp[i], q->wm[i]);
            if (fabs(y) > cap) {
                double sc = cap / fabs(y);
                q->vl[i] *= sc; q->vu[i] *= sc; y *= sc;
            }
        } else if (y_start) {            /* equality row, warm start: the previous multiplier, inside the penalty box */
            double cap = 0.5 * fmin(q->wp[i], q->wm[i]);
            y = fmax(-cap, fmin(cap, y_start[i]));
        }
        q->s[i] = s; q->y[i] = y;
        q->zp[i] = q->wp[i] - y; q->zm[i] = q->wm[i] + y;
        q->tp[i] = fmax(d, 0.0) + mu0 / (q->wp[i] - y);
        q->tm[i] = fmax(-d, 0.0) + mu0 / (q->wm[i] + y);
        /* keep the row equation exact: tp - tm = d */
        double e = (q->tp[i] - q->tm[i]) - d;
        if (e > 0) q->tm[i] += e; else q->tp[i] -= e;
    }
}

/* Monotone (Fiacco-McCormick) barrier method with a primal-feasible interior start (the elastic
 * variables absorb every row residual).  Constants follow the published Ipopt defaults
 * (kappa_eps=10, kappa_mu=0.2, theta_mu=1.5, tau_min=0.99).
 * returns 0 converged, 1 iteration limit, 2 numerical failure */
static int ipm_run(ora_qp *q, const double *p_start, const double *y_start)
{
    int64_t n = q->n, m = q->m;
    double *rd = (double *)malloc(sizeof(double) * (size_t)(n + 1));
    double *rp = (double *)malloc(sizeof(double) * (size_t)(m + 1));
    ipm_init(q, p_start, y_start);
    q->delta_w_last = 0.0;
    int rc = 1;
    const double tol = q->opt.ipm_tol;
    const double mu_min = tol / 10.0;
    double mu = 1.0;
    int n_acc = 0, n_acc2 = 0, n_acc3 = 0;
    double dw_prev = 0.0;
    /* predictor-corrector mode (opt.ipm_corrector) until the first inertia correction of this solve */
    int mpc = q->opt.ipm_corrector != 0;
    double *soc = mpc ? (double *)calloc((size_t)(2 * n + 4 * m + 1), sizeof(double)) : NULL;
    int verbose = getenv("ORA_IPM_VERBOSE") != NULL;
    const int eq_steps = getenv("ORA_EQ_STEPS") ? atoi(getenv("ORA_EQ_STEPS")) : 0;
    int short_lim = 0, n_short = 0;
    double short_a = 0.1;
    double short_dw = 0.0;
    if (getenv("ORA_MPC_SHORT")) sscanf(getenv("ORA_MPC_SHORT"), "%d,%lf,%lf", &short_lim, &short_a, &short_dw);
    /* iteration limit; half of it for a second-order correction: one that has not converged by then is abandoned,
     * which for run! is the same as any other unsuccessful correction (same rule as the product, ipm.hip k_ipm_prepare) */
    const int it_max = q->cur_mode == ORA_MODE_SOC ? q->opt.ipm_max_iter / 2 : q->opt.ipm_max_iter;
    for (int it = 0; it < it_max; ++it) {
        ipm_meas ms;
        ipm_measure(q, rd, rp, &ms);
        if (!isfinite(ms.rd) || !isfinite(ms.cavg) || !isfinite(ms.rp)) { rc = 2; break; }
        double sd = fmax(100.0, ms.dual_l1 / (double)(n + m)) / 100.0;
        double e0 = fmax(fmax(ms.rd / sd, ms.rp), ms.cmax / sd);
        q->last_e0 = e0; q->last_rule = -1;
        if (e0 <= tol) { rc = 0; q->last_rule = 0; break; }
        /* acceptable termination: 8 consecutive iterates within 100 x tol (the monotone rule can crawl
         * just above the tolerance when round-off keeps triggering tiny inertia corrections) */
        n_acc = e0 <= 100.0 * tol ? n_acc + 1 : 0;
        /* ... or 15 within 1000 x tol (Ipopt's acceptable_tol / acceptable_iter ratio): the fixed 1e-8 dual
         * regularisation leaves a row residual of 1e-8 |dy| that full Newton steps cannot remove when the multiplier
         * steps stay large (seen at trust-region radii ~1e-5: rp stalls at 1.9e-7) */
        n_acc2 = e0 <= 1000.0 * tol ? n_acc2 + 1 : 0;
        /* ... or 25 within 10^4 x tol (round 3): on the 9241-bus shape the regularised Newton iteration of a sub-problem with
         * nearly flat directions (delta_w 1e-4 ... 1e-3 against curvatures of 1e-6) stalls at an error of 1e-6 ... 4e-6 for as
         * long as it is allowed to -- 98 of 128 line-outage scenarios ran their second trust-region QP into the
         * 200-iteration limit and run! gave up on them.  The rule never fires on the IEEE-118 bench workload (1 282
         * sub-problems: same iteration and factorisation counts with and without it); same rule in ipm.hip (b_ipm_prepare) */
        n_acc3 = e0 <= 1e4 * tol ? n_acc3 + 1 : 0;
        if (n_acc >= 8 || n_acc2 >= 15 || n_acc3 >= 25) { rc = 0; q->last_rule = n_acc >= 8 ? 1 : (n_acc2 >= 15 ? 2 : 3); break; }
        /* barrier update */
        for (int k = 0; k < 20 && !mpc; ++k) {
            double emu = fmax(fmax(ms.rd / sd, ms.rp), ipm_compl_err(q, mu) / sd);
            if (emu > 10.0 * mu || mu <= mu_min) break;
            mu = fmax(mu_min, fmin(0.2 * mu, pow(mu, 1.5)));
        }
        double tau = fmax(0.99, 1.0 - mu);
        q->ipm_iters++;
        for (int64_t j = 0; j < n; ++j) {
            double sg = 0.0;
            if (isfinite(q->lb[j])) sg += q->zl[j] / (q->p[j] - q->lb[j]);
            if (isfinite(q->ub[j])) sg += q->zu[j] / (q->ub[j] - q->p[j]);
            q->sigp[j] = sg;
        }
        for (int64_t i = 0; i < m; ++i) {
            if (q->rtype[i] == ROW_FREE) { q->D[i] = 1.0; continue; }
            double d = q->tp[i] / q->zp[i] + q->tm[i] / q->zm[i];
            if (q->rtype[i] == ROW_INEQ) {
                double sig = 0.0;
                if (isfinite(q->lo[i])) sig += q->vl[i] / (q->s[i] - q->lo[i]);
                if (isfinite(q->hi[i])) sig += q->vu[i] / (q->hi[i] - q->s[i]);
                d += 1.0 / sig;
            }
            q->D[i] = d;
        }
        /* Newton direction of the regularised system; a larger delta_w is tried if the solve is
         * inaccurate.  Steps are the fraction-to-boundary lengths (primal and dual separately). */
        double dw = 0.0, alpha = 0.0, a_d = 0.0, relres = 0.0;
        int ok = 0;
        /* if the previous iteration of this solve needed an inertia correction, the zero trial is skipped
         * and a third of that correction is tried first (it decays geometrically while it keeps working) */
        double dw_floor = dw_prev > 3e-10 ? fmax(1e-20, dw_prev / 3.0) : 0.0;
        for (int attempt = 0; attempt < 12 && !ok; ++attempt) {
            if (kkt_factor(q, dw_floor, &dw) != 0) break;
            double apm, adm;
            if (mpc && dw > 0.0) {
                /* first inertia correction: the sub-problem is not convex along the path; from here on the
                 * monotone rule, restarted from the current average complementarity */
                mpc = 0;
                mu = fmax(mu_min, fmin(1.0, ms.cavg));
                tau = fmax(0.99, 1.0 - mu);
            }
            if (mpc) {
                /* predictor: affine-scaling direction (target 0), largest steps to the boundary */
                relres = ipm_direction(q, dw, 0.0, rd, rp, NULL, 1);
                ipm_max_steps(q, &apm, &adm);
                const double ap1 = fmin(1.0, apm), ad1 = fmin(1.0, adm);
                double csum = 0.0;
                int64_t nc = 0;
#define PR(zz, dz, xx, dx, slot) do { csum += ((zz) + ad1 * (dz)) * ((xx) + ap1 * (dx)); ++nc; soc[slot] = (dz) * (dx); } while (0)
                for (int64_t j = 0; j < n; ++j) {
                    soc[j] = 0.0; soc[n + j] = 0.0;
                    if (isfinite(q->lb[j])) PR(q->zl[j], q->dzl[j], q->p[j] - q->lb[j], q->dp[j], j);
                    if (isfinite(q->ub[j])) PR(q->zu[j], q->dzu[j], q->ub[j] - q->p[j], -q->dp[j], n + j);
                }
                for (int64_t i = 0; i < m; ++i) {
                    soc[2 * n + i] = soc[2 * n + m + i] = soc[2 * n + 2 * m + i] = soc[2 * n + 3 * m + i] = 0.0;
                    if (q->rtype[i] == ROW_FREE) continue;
                    PR(q->zp[i], -q->dy[i], q->tp[i], q->dtp[i], 2 * n + i);
                    PR(q->zm[i], q->dy[i], q->tm[i], q->dtm[i], 2 * n + m + i);
                    if (q->rtype[i] == ROW_INEQ) {
                        if (isfinite(q->lo[i])) PR(q->vl[i], q->dvl[i], q->s[i] - q->lo[i], q->ds[i], 2 * n + 2 * m + i);
                        if (isfinite(q->hi[i])) PR(q->vu[i], q->dvu[i], q->hi[i] - q->s[i], -q->ds[i], 2 * n + 3 * m + i);
                    }
                }
#undef PR
                /* Mehrotra's centring parameter sigma = (mu_aff / mu_now)^3, clamped to [1e-4, 1] */
                const double mu_aff = nc ? csum / (double)nc : 0.0;
                double sigma = ms.cavg > 0.0 ? pow(fmax(0.0, mu_aff) / ms.cavg, 3.0) : 1.0;
                sigma = fmin(1.0, fmax(sigma, 1e-4));
                mu = fmax(mu_min, sigma * ms.cavg);
                tau = fmax(0.99, 1.0 - mu);
                /* corrector: centring target mu with the second-order terms, same factorisation */
                relres = ipm_direction(q, dw, mu, rd, rp, soc, 0);
            } else {
                relres = ipm_direction(q, dw, mu, rd, rp, NULL, 0);
            }
            ipm_max_steps(q, &apm, &adm);
            alpha = fmin(1.0, tau * apm);
            a_d = fmin(1.0, tau * adm);
            if (eq_steps == 1 || (eq_steps == 2 && !mpc) || (eq_steps == 3 && dw > 0.0)) alpha = a_d = fmin(alpha, a_d);   /* experiment */
            if (isfinite(relres) && relres < 1e-6 && isfinite(alpha) && isfinite(a_d)) { ok = 1; break; }
            dw_floor = dw > 0.0 ? 8.0 * dw : (q->delta_w_last > 0.0 ? q->delta_w_last : 1e-4);
            if (dw_floor > 1e20) break;
        }
        if (!ok) { rc = 2; break; }
        dw_prev = dw;
        if (mpc && short_lim > 0) {     /* experiment (ORA_MPC_SHORT=k,a[,dw]): k consecutive predictor-corrector steps below a end that mode
                                         * (and, with dw, the next iteration starts from that regularisation: a Levenberg-Marquardt damping) */
            n_short = alpha < short_a ? n_short + 1 : 0;
            if (n_short >= short_lim) { mpc = 0; mu = fmax(mu_min, fmin(1.0, ms.cavg)); if (short_dw > 0.0) dw_prev = 3.0 * short_dw; }
        }
        for (int64_t j = 0; j < n; ++j) {
            q->p[j] = nudge_inside(q->p[j] + alpha * q->dp[j], q->lb[j], q->ub[j]);
            q->zl[j] += a_d * q->dzl[j];
            q->zu[j] += a_d * q->dzu[j];
        }
        for (int64_t i = 0; i < m; ++i) {
            if (q->rtype[i] == ROW_FREE) continue;
            q->tp[i] += alpha * q->dtp[i]; q->tm[i] += alpha * q->dtm[i];
            q->s[i] += alpha * q->ds[i];
            if (q->rtype[i] == ROW_INEQ) q->s[i] = nudge_inside(q->s[i], q->lo[i], q->hi[i]);
            q->vl[i] += a_d * q->dvl[i]; q->vu[i] += a_d * q->dvu[i];
            if (q->rtype[i] == ROW_INEQ) q->y[i] = q->vl[i] - q->vu[i];
            else q->y[i] += a_d * q->dy[i];
            q->zp[i] -= a_d * q->dy[i]; q->zm[i] += a_d * q->dy[i];
        }
        if (verbose)
            fprintf(stderr, "  ipm %3d mu=%.2e e0=%.2e rd=%.2e rp=%.1e cmax=%.2e a=%.3f ad=%.3f dw=%.1e rr=%.1e\n",
                    it, mu, e0, ms.rd, ms.rp, ms.cmax, alpha, a_d, dw, relres);
    }
    free(rd); free(rp); free(soc);
    return rc;
}

/* ------------------------------------------------------------------ mode setup + reporting */
#define RHO_BIG0 1e4
#define RHO_BIG_MAX 1e10
#define ELASTIC_TOL 1e-8
#define RHO_CERT_FRAC 0.5

static void set_weights(ora_qp *q, double rho_big, double soft_w, int phase1)
{
    for (int64_t i = 0; i < q->m; ++i) {
        if (q->rtype[i] == ROW_FREE) continue;
        if (q->hard[i]) { q->wp[i] = q->wm[i] = phase1 ? 1.0 : rho_big; }
        else { q->wp[i] = q->wm[i] = soft_w; }
    }
}

static double hard_elastic(const ora_qp *q)
{
    double e = 0.0;
    for (int64_t i = 0; i < q->m; ++i)
        if (q->rtype[i] != ROW_FREE && q->hard[i]) e = fmax(e, fmax(q->tp[i], q->tm[i]));
    return e;
}

int ora_qp_solve(ora_qp *q, int mode, const double *x_k, double delta, double mu_pen,
                 const double *c, const double *b, const double *jval, const double *hval,
                 double *p, double *lambda, double *mult_x_U, double *mult_x_L, double *slack)
{
    int64_t n = q->n, m = q->m;
    q->ipm_iters = 0; q->n_factor = 0; q->last_elastic = 0.0;
    q->cur_mode = mode;
    memcpy(q->jv, jval, sizeof(double) * (size_t)q->nnzj);
    int use_obj = (mode == ORA_MODE_QP || mode == ORA_MODE_SOC || mode == ORA_MODE_L1QP);
    double *pstart = NULL;
    if (mode == ORA_MODE_LP) {
        /* subproblem_JuMP.jl:185-244: min sum (x - x_k)^2 over absolute x, linear rows only */
        for (int64_t j = 0; j < n; ++j) {
            q->lb[j] = q->xL[j]; q->ub[j] = q->xU[j];
            q->c[j] = -2.0 * x_k[j]; q->hd[j] = 2.0;
        }
        memset(q->hv, 0, sizeof(double) * (size_t)q->nnzh);
        pstart = (double *)malloc(sizeof(double) * (size_t)n);
        memcpy(pstart, x_k, sizeof(double) * (size_t)n);
    } else {
        /* subproblem_JuMP.jl:432-456 */
        for (int64_t j = 0; j < n; ++j) {
            double vl = q->xL[j] - x_k[j], vu = q->xU[j] - x_k[j];
            double lb = fmax(-delta, vl), ub = fmin(delta, vu);
            if (lb > ub) { lb = fmax(-delta, fmin(0.0, vl)); ub = fmin(delta, fmax(0.0, vu)); }
            q->lb[j] = lb; q->ub[j] = ub;
            q->c[j] = use_obj ? c[j] : 0.0;
            q->hd[j] = 0.0;
        }
        if (use_obj && hval && q->nnzh) memcpy(q->hv, hval, sizeof(double) * (size_t)q->nnzh);
        else memset(q->hv, 0, sizeof(double) * (size_t)q->nnzh);
    }
    /* degenerate boxes get a hair of interior */
    for (int64_t j = 0; j < n; ++j)
        if (isfinite(q->lb[j]) && isfinite(q->ub[j]) && q->ub[j] - q->lb[j] < 1e-8) {
            double mid = 0.5 * (q->lb[j] + q->ub[j]);
            q->lb[j] = mid - 5e-9; q->ub[j] = mid + 5e-9;
        }
    /* rows: subproblem_JuMP.jl:79-112 (typing), :492-505 (shifted bounds) */
    for (int64_t i = 0; i < m; ++i) {
        double gl = q->gL[i], gu = q->gU[i];
        if (mode == ORA_MODE_LP) {
            if (i >= q->nlin) { q->rtype[i] = ROW_FREE; continue; }
            q->lo[i] = gl; q->hi[i] = gu;
        } else {
            q->lo[i] = gl - b[i]; q->hi[i] = gu - b[i];
        }
        if (gl == gu) q->rtype[i] = ROW_EQ;
        else if (gl > -INFINITY || gu < INFINITY) q->rtype[i] = ROW_INEQ;
        else q->rtype[i] = ROW_FREE;
        int nonlinear = i >= q->nlin;
        switch (mode) {
        case ORA_MODE_FR:      /* subproblem_JuMP.jl:365-380 */
            q->hard[i] = !(nonlinear && !(b[i] >= gl && b[i] <= gu));
            break;
        case ORA_MODE_L1QP:    /* :324-330 */
        case ORA_MODE_INFEAS:  /* :407-413 */
            q->hard[i] = !nonlinear;
            break;
        default:
            q->hard[i] = 1;
        }
    }
    /* objective scaling (external-solver internals, not reference semantics) */
    double cmax = 0.0;
    for (int64_t j = 0; j < n; ++j) cmax = fmax(cmax, fabs(q->c[j]));
    q->sf = cmax > 100.0 ? 100.0 / cmax : 1.0;
    if (q->sf != 1.0) {
        for (int64_t j = 0; j < n; ++j) { q->c[j] *= q->sf; q->hd[j] *= q->sf; }
        for (int64_t k = 0; k < q->nnzh; ++k) q->hv[k] *= q->sf;
    }
    double soft_w = (mode == ORA_MODE_L1QP ? mu_pen : 1.0) * q->sf;

    /* warm start: the first run starts from the step and equality multipliers of the previous solved sub-problem of
     * the same mode (still in q->p / q->y); restarts with a larger penalty start cold */
    double *ystart = NULL;
    if (q->opt.ipm_warm_start && mode != ORA_MODE_LP && q->prev_mode == mode) {
        pstart = (double *)malloc(sizeof(double) * (size_t)n);
        memcpy(pstart, q->p, sizeof(double) * (size_t)n);
        ystart = (double *)malloc(sizeof(double) * (size_t)(m + 1));
        for (int64_t i = 0; i < m; ++i) ystart[i] = q->y[i] * (q->sf / q->prev_sf);   /* into this solve's objective scale */
    }
    int status = ORA_MOI_OTHER_ERROR;
    double rho_big = RHO_BIG0;
    for (int run = 0;; ++run) {
        set_weights(q, rho_big, soft_w, 0);
        if (run > 0 && mode != ORA_MODE_LP) { free(pstart); pstart = NULL; free(ystart); ystart = NULL; }
        int rc = ipm_run(q, pstart, ystart);
        if (rc == 1) { status = ORA_MOI_ITERATION_LIMIT; break; }
        if (rc == 2) { status = ORA_MOI_NUMERICAL_ERROR; break; }
        q->last_elastic = hard_elastic(q);
        if (q->last_elastic <= ELASTIC_TOL) { status = ORA_MOI_LOCALLY_SOLVED; break; }
        if (!q->opt.ipm_phase1) {
            /* elastic mass on a hard row: infeasible, or penalty too small?  H p + c = J'y + zl - zu; for an infeasible
             * programme the terms of the right-hand side (size rho) cancel down to the objective gradient (the scaled
             * multipliers are an infeasibility certificate); no cancellation means the penalty only balances the
             * objective: raise it and solve again (same rule as the product, ipm.hip k_qp_finish) */
            double *g = (double *)malloc(sizeof(double) * (size_t)(n + 1));
            hess_mul(q, NULL, q->p, g);
            double gm = 0.0, am = 0.0;
            for (int64_t j = 0; j < n; ++j) {
                gm = fmax(gm, fabs(g[j] + q->c[j]));
                double t = (isfinite(q->lb[j]) ? q->zl[j] : 0.0) + (isfinite(q->ub[j]) ? q->zu[j] : 0.0);
                for (int64_t k = q->jcolptr[j]; k < q->jcolptr[j + 1]; ++k) {
                    int64_t i = q->jrowval[k];
                    if (q->rtype[i] != ROW_FREE) t += fabs(q->jv[k] * q->y[i]);
                }
                am = fmax(am, t);
            }
            free(g);
            if (gm > RHO_CERT_FRAC * am && rho_big < RHO_BIG_MAX) { rho_big *= 100.0; continue; }
            status = ORA_MOI_LOCALLY_INFEASIBLE; break;
        }
        /* elastic mass on a hard row: infeasible, or penalty too small?  Phase 1 decides. */
        {
            size_t nb = sizeof(double) * (size_t)n, mb = sizeof(double) * (size_t)m;
            double *sv = (double *)malloc(3 * nb + 6 * mb + nb + sizeof(double) * (size_t)(q->nnzh + 1));
            double *sp = sv, *szl = sp + n, *szu = szl + n, *ss = szu + n, *stp = ss + m, *stm = stp + m,
                   *sy = stm + m, *svl = sy + m, *svu = svl + m, *sc = svu + m, *shv = sc + n;
            memcpy(sp, q->p, nb); memcpy(szl, q->zl, nb); memcpy(szu, q->zu, nb);
            memcpy(ss, q->s, mb); memcpy(stp, q->tp, mb); memcpy(stm, q->tm, mb);
            memcpy(sy, q->y, mb); memcpy(svl, q->vl, mb); memcpy(svu, q->vu, mb);
            memcpy(sc, q->c, nb); memcpy(shv, q->hv, sizeof(double) * (size_t)q->nnzh);
            int *srt = (int *)malloc(sizeof(int) * (size_t)(m + 1));
            memcpy(srt, q->rtype, sizeof(int) * (size_t)m);
            double *shd = (double *)malloc(nb + 8);
            memcpy(shd, q->hd, nb);
            memset(q->c, 0, nb); memset(q->hv, 0, sizeof(double) * (size_t)q->nnzh);
            memset(q->hd, 0, nb);
            for (int64_t i = 0; i < m; ++i) if (!q->hard[i]) q->rtype[i] = ROW_FREE;
            set_weights(q, 1.0, 1.0, 1);
            int rc1 = ipm_run(q, pstart, NULL);
            double e1 = hard_elastic(q);
            memcpy(q->rtype, srt, sizeof(int) * (size_t)m);
            memcpy(q->c, sc, nb); memcpy(q->hv, shv, sizeof(double) * (size_t)q->nnzh);
            memcpy(q->hd, shd, nb);
            int infeasible = (rc1 != 0) || e1 > ELASTIC_TOL;
            if (infeasible || rho_big >= RHO_BIG_MAX) {
                memcpy(q->p, sp, nb); memcpy(q->zl, szl, nb); memcpy(q->zu, szu, nb);
                memcpy(q->s, ss, mb); memcpy(q->tp, stp, mb); memcpy(q->tm, stm, mb);
                memcpy(q->y, sy, mb); memcpy(q->vl, svl, mb); memcpy(q->vu, svu, mb);
                free(sv); free(srt); free(shd);
                status = ORA_MOI_LOCALLY_INFEASIBLE;
                break;
            }
            free(sv); free(srt); free(shd);
            rho_big *= 100.0;
        }
    }
    free(pstart); free(ystart);
    q->prev_mode = status == ORA_MOI_LOCALLY_SOLVED ? mode : -1;
    q->prev_sf = q->sf;

    /* collect_solution!: subproblem_JuMP.jl:514-563 */
    if (status == ORA_MOI_LOCALLY_SOLVED) {
        for (int64_t j = 0; j < n; ++j) {
            p[j] = q->p[j];
            double rc = ((isfinite(q->lb[j]) ? q->zl[j] : 0.0) - (isfinite(q->ub[j]) ? q->zu[j] : 0.0)) / q->sf;
            mult_x_L[j] = 0.0; mult_x_U[j] = 0.0;
            if (rc > 0) mult_x_L[j] = rc; else if (rc < 0) mult_x_U[j] = rc;
        }
        for (int64_t i = 0; i < m; ++i) lambda[i] = q->rtype[i] == ROW_FREE ? 0.0 : q->y[i] / q->sf;
        if (slack) for (int64_t i = 0; i < m; ++i) { slack[i] = q->tp[i]; slack[m + i] = q->tm[i]; }
    } else {
        /* infeasible family -> zeros (:551-555); other statuses leave the reference's outputs
         * undefined (:556-560), zeros here */
        for (int64_t j = 0; j < n; ++j) { p[j] = 0.0; mult_x_L[j] = 0.0; mult_x_U[j] = 0.0; }
        for (int64_t i = 0; i < m; ++i) lambda[i] = 0.0;
        if (slack) memset(slack, 0, sizeof(double) * (size_t)(2 * m));
    }
    return status;
}
