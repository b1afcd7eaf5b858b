/*
 * sqp_oracle.h -- CPU restatement of the SqpSolver.jl hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (libsqphip.so) never links, loads or calls anything in oracle/.
 *
 * What is restated, function by function (citations into /root/reference):
 *   ora_norm_violations      src/algorithms/common.jl:54-77
 *   ora_kt_residuals         src/algorithms/common.jl:14-23
 *   ora_norm_complementarity src/algorithms/common.jl:30-47
 *   ora_compute_derivative   src/algorithms/merit.jl:13-17, src/algorithms/sqp.jl:190-213
 *   ora_sqp_tr_solve         src/algorithms/sqp_trust_region.jl:26-91 (state), :98-223 (run!),
 *                            :237-304 (linear phase), :314-380 (sub_optimize!/SOC/compute_step!),
 *                            :487-579 (compute_qmodel, do_step!); src/algorithms/sqp.jl:66-224
 *   ora_qp_solve             src/algorithms/subproblem_JuMP.jl:36-125 (row typing), :127-183 (QP),
 *                            :185-244 (LP phase), :283-347 (L1QP), :352-393 (FR), :398-429 (INFEAS),
 *                            :432-463 (trust region bounds), :514-563 (collect_solution!)
 *   ora_armijo_alpha         src/algorithms/sqp_line_search.jl:303-334 (dead code upstream; semantics only)
 *
 * PARITY STATUS: "parity unpinned" for the QP arithmetic itself.  The reference delegates every
 * QP to an external solver (Ipopt + MUMPS/MA57, not vendored, version unpinned:
 * test/Project.toml:8, examples/acopf/opf.jl:59-64) and neither Julia nor Ipopt exists in this
 * image, so the reference cannot be run.  The interior-point method in qp_ipm.c plays Ipopt's
 * role; it is pinned only by the reference's own known answers (test/runtests.jl:12-14,
 * README.md:18-21) and by hand-derived KKT conditions, see tests/test_oracle_kat.py.
 */
#ifndef SQP_ORACLE_H
#define SQP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* MOI.TerminationStatusCode integers (MathOptInterface v1 enum order; MOI is not vendored,
 * call sites: subproblem_JuMP.jl:179,525,551,556; sqp_trust_region.jl:144,151) */
enum {
    ORA_MOI_OPTIMIZE_NOT_CALLED = 0, ORA_MOI_OPTIMAL = 1, ORA_MOI_INFEASIBLE = 2,
    ORA_MOI_DUAL_INFEASIBLE = 3, ORA_MOI_LOCALLY_SOLVED = 4, ORA_MOI_LOCALLY_INFEASIBLE = 5,
    ORA_MOI_ALMOST_OPTIMAL = 7, ORA_MOI_ALMOST_LOCALLY_SOLVED = 10,
    ORA_MOI_ITERATION_LIMIT = 11, ORA_MOI_NUMERICAL_ERROR = 20, ORA_MOI_OTHER_ERROR = 24
};

/* sub-problem modes */
enum { ORA_MODE_QP = 0, ORA_MODE_FR = 1, ORA_MODE_SOC = 2, ORA_MODE_LP = 3,
       ORA_MODE_L1QP = 4, ORA_MODE_INFEAS = 5 };

/* src/parameters.jl:17-29 defaults via ora_default_options */
typedef struct {
    double tol_direction, tol_residual, tol_infeas;
    double init_mu, max_mu, tr_size;
    double rho, eta, tau, min_alpha;
    int max_iter, use_soc;
    int literal_quirks; /* 1 (default): reproduce SURVEY.md App. C quirks #2/#3 (Hessian and KT residual
                           built from JuMP-sign multipliers); 0: textbook signs */
    /* stand-in for the external solver's own options */
    double ipm_tol;
    int ipm_max_iter;
    int ipm_phase1;    /* 1: confirm an infeasibility verdict with a phase-1 run and escalate the penalty if it
                          disagrees; 0 (default): elastic mass left on a hard row means infeasible */
    int num_threads;   /* OpenMP threads for the dense LDL^T (cpu_baseline reports this) */
    int ipm_corrector; /* 0 (default since round 4): monotone Fiacco-McCormick rule throughout, Ipopt's default barrier strategy
                          (mu_strategy = monotone; the reference's tests and examples leave it there); 1: Mehrotra
                          predictor-corrector (adaptive barrier parameter + second-order term, two solves per factorisation)
                          while the sub-problem behaves convex, monotone rule from the first inertia correction on */
    int kkt_condense;  /* 1: eliminate the rows with gL != gU (their block of the Newton matrix is the diagonal -D)
                          before factorising: dense LDL^T of order n + #equality rows instead of n + m */
    int kkt_tile_order; /* 0 (default): the oracle orders its factorisation by itself.  1 (tests of the product's dense
                           tile order only): factorise the condensed matrix in the order the product library uses (its
                           host-only sqphip_kkt_order), so that both sides stay on one rounding trajectory; the
                           permutation is handed in by oracle.py through ora_set_kkt_order -- the C code itself does
                           not read this field */
    int kkt_mode;       /* linear algebra of the Newton systems: 1 dense LDL^T (above), 2 sparse LDL^T with the oracle's
                           own minimum-degree order (sparse_ldlt.c), 0 (default): sparse from order 3000 up, else dense */
    int ipm_warm_start; /* 1: the first interior-point run of a sub-problem starts from the step and the equality-row
                           multipliers of the previous solved sub-problem of the same mode (the reference passes
                           warm_start_init_point = "yes" to Ipopt, test/ext_solver.jl:5); 0 (default): cold start */
} ora_options;

/* sparse_ldlt.c: sparse LDL^T without pivoting (ordering + up-looking factorisation), the checker of the product's
 * multifrontal path.  Triplets (ti, tj) in any triangle, duplicates summed; bp / bi (may be NULL): CSR lists of the
 * indices that must be eliminated before index u; natural != 0 skips the ordering. */
typedef struct ora_sldl ora_sldl;
ora_sldl *ora_sldl_analyse(int64_t n, int64_t nt, const int64_t *ti, const int64_t *tj, const int64_t *bp,
                           const int64_t *bi, int natural);
void ora_sldl_free(ora_sldl *S);
int64_t ora_sldl_nnz_l(const ora_sldl *S);
const int64_t *ora_sldl_perm(const ora_sldl *S);
const double *ora_sldl_pivots(const ora_sldl *S);   /* D of the last numeric factorisation, elimination order */
int64_t ora_sldl_numeric(ora_sldl *S, const double *tv, int64_t *nbad);
void ora_sldl_solve(const ora_sldl *S, double *x);

typedef double (*ora_eval_f_t)(void *ud, const double *x);
typedef void (*ora_eval_grad_f_t)(void *ud, const double *x, double *grad);
typedef void (*ora_eval_g_t)(void *ud, const double *x, double *g);
typedef void (*ora_eval_jac_g_t)(void *ud, const double *x, double *vals);
typedef void (*ora_eval_h_t)(void *ud, const double *x, double obj_factor,
                             const double *lambda, double *vals);

/* Mirror of SqpSolver.Model (src/model.jl:3-35) */
typedef struct {
    int64_t n, m, num_linear;
    int64_t nnzj, nnzh;
    const int64_t *jrow, *jcol;   /* 1-based COO, duplicates allowed */
    const int64_t *hrow, *hcol;   /* 1-based triangular COO, duplicates allowed; nnzh==0 => no Hessian */
    const double *xL, *xU, *gL, *gU;
    ora_eval_f_t eval_f;
    ora_eval_grad_f_t eval_grad_f;
    ora_eval_g_t eval_g;
    ora_eval_jac_g_t eval_jac_g;
    ora_eval_h_t eval_h;          /* may be NULL */
    void *ud;
} ora_nlp;

/* Per-iteration trace row (the columns the reference prints, sqp_trust_region.jl:605-634,
 * plus the discrete decisions needed for trace parity) */
typedef struct {
    int iter;
    int accepted;       /* step_acceptance at print time */
    int fr;             /* feasibility_restoration at print time */
    int sub_status;     /* MOI code of the QP solve of this iteration */
    int ipm_iters;      /* interior-point iterations spent in this iteration's sub-solves */
    int n_factor;       /* KKT factorisations spent */
    double f, phi, mu, delta, pnorm, prim_infeas, dual_infeas;
} ora_trace_row;

typedef struct {
    int status;         /* src/status.jl codes, as problem.status (sqp_trust_region.jl:216) */
    int iter;           /* sqp.iter on exit (sqp_trust_region.jl:222) */
    double obj_val;
    int n_qp;           /* number of sub-problem solves (QP/FR/SOC/LP) */
    int n_ipm_iter;     /* total interior-point iterations */
    int n_factor;       /* total KKT factorisations */
    double qp_seconds;  /* wall time in sub-problem solves */
    int trace_len;
} ora_result;

void ora_default_options(ora_options *o);

/* ---- R1..R4, M2: scalar reductions ------------------------------------------------------- */
/* pnorm: 1, 2 or 0 meaning Inf */
double ora_norm_violations(int64_t m, int64_t n, const double *E, const double *gL,
                           const double *gU, const double *x, const double *xL,
                           const double *xU, int pnorm);
/* Jacobian given as CSC (colptr[n+1], rowval 0-based, nzval) */
double ora_kt_residuals(int64_t m, int64_t n, const double *df, const double *lambda,
                        const double *mult_x_U, const double *mult_x_L,
                        const int64_t *colptr, const int64_t *rowval, const double *nzval);
double ora_norm_complementarity(int64_t m, const double *E, const double *gL, const double *gU,
                                const double *lambda, int pnorm);
double ora_compute_derivative(double dfp, double mu, int64_t m, const double *cons_viol);
double ora_compute_derivative_full(int64_t n, int64_t m, const double *df, const double *p, const double *E,
                                   const double *gL, const double *gU, double mu, const double *mu_vec,
                                   int feasibility_restoration, const double *slack, int64_t nslack);
void ora_compute_mu_rule(int rule, int64_t iter, double rho, double viol1, double dfp, double half_pHp, int64_t m,
                         const double *lambda, double *mu);
int ora_isapprox(double a, double b);

/* ---- the QP sub-problem (Ipopt's seat) ---------------------------------------------------- */
typedef struct ora_qp ora_qp;
/* J and H patterns are CSC, 0-based; H holds BOTH triangles (sqp.jl:96-101) or nnz 0 */
ora_qp *ora_qp_create(int64_t n, int64_t m, int64_t num_linear,
                      const int64_t *jcolptr, const int64_t *jrowval,
                      const int64_t *hcolptr, const int64_t *hrowval,
                      const double *xL, const double *xU, const double *gL, const double *gU,
                      const ora_options *opt);
void ora_qp_destroy(ora_qp *qp);
/* Returns MOI status. Outputs in JuMP sign convention (collect_solution!, :514-563).
 * slack may be NULL, else 2*m (t+ then t-). For ORA_MODE_LP `p` receives the absolute x. */
int ora_qp_solve(ora_qp *qp, int mode, const double *x_k, double delta, double mu_pen,
                 const double *c, const double *b, const double *jval, const double *hval,
                 double *p, double *lambda, double *mult_x_U, double *mult_x_L, double *slack);
/* problems.c: drop the second derivatives of a problem (the reference's eval_h === nothing path: SLP) */
struct ora_problem;
void ora_problem_drop_hessian(struct ora_problem *p);
void ora_qp_termination(const ora_qp *q, int *rule, double *scaled_error);
void ora_qp_stats(const ora_qp *qp, int *ipm_iters, int *n_factor, double *last_elastic);

/* ---- dense LDL^T, exported for kernel-level parity tests and the CPU baseline -------------- */
/* In-place on column-major lower triangle, leading dimension ld. No pivoting.
 * dinv receives 1/D. Returns the number of positive pivots among the first n1 rows in *npos1
 * and negative pivots among the rest in *nneg2. */
void ora_ldlt_factor(int64_t N, double *A, int64_t ld, double *dinv, int64_t n1,
                     int64_t *npos1, int64_t *nneg2, int nthreads);
void ora_ldlt_solve(int64_t N, const double *A, int64_t ld, const double *dinv, double *x);

/* ---- SQP-TR driver ------------------------------------------------------------------------ */
/* x: in = start point, out = solution.  trace may be NULL (capacity trace_cap rows). */
void ora_sqp_tr_solve(const ora_nlp *nlp, const ora_options *opt, double *x, double *g,
                      double *mult_g, double *mult_x_L, double *mult_x_U,
                      ora_result *res, ora_trace_row *trace, int trace_cap);

/* Armijo backtracking of the (dead) line-search variant: returns alpha, *valid = 0 on failure.
 * phi_at(alpha) callback evaluates the merit at x + alpha p. */
double ora_armijo_alpha(double phi0, double dir_deriv, double pnorm_inf, double tol_direction,
                        double eta, double tau, double min_alpha,
                        double (*phi_at)(void *, double), void *ud, int *valid);

/* ---- built-in test problems ---------------------------------------------------------------- */
/* toy: test/ext_solver.jl:14-28; readme1: README.md:18-21; hs071 (MOI.Test, not vendored) */
typedef struct ora_problem ora_problem;
ora_problem *ora_problem_toy(void);
ora_problem *ora_problem_readme1(void);
ora_problem *ora_problem_hs071(void);
/* test hook: order of the condensed Newton matrix for every QP solver created afterwards (len 0 = natural) */
void ora_set_kkt_order(const int32_t *rank, int64_t len);
/* ACOPF evaluator over the arrays of sqpsolver.jl_amd/acopf_synth.py (Network + NlpLayout) */
ora_problem *ora_problem_acopf(int nb, int ng, int nl, const int32_t *f_bus,
                               const int32_t *t_bus, const double *ohm /* [nl][12] */,
                               const int32_t *gen_bus, const double *c2,
                               const double *c1, const int32_t *bal_ptr, const int32_t *bal_colP,
                               const int32_t *bal_colQ, const double *bal_coef,
                               int64_t nnzj, const int64_t *jrow, const int64_t *jcol,
                               int64_t nnzh, const int64_t *hrow, const int64_t *hcol,
                               const double *xL, const double *xU, const double *gL,
                               const double *gU, int nsh, const int32_t *sh_bus, const double *sh_gs,
                               const double *sh_bs, int ndc, const double *dc_loss1);
/* the same network in rectangular voltage coordinates (ACRPowerModel; acopf_synth.py, acr_layout) */
ora_problem *ora_problem_acopf_acr(int nb, int ng, int nl, const int32_t *f_bus,
                               const int32_t *t_bus, const double *ohm /* [nl][12] */,
                               const int32_t *gen_bus, const double *c2,
                               const double *c1, const int32_t *bal_ptr, const int32_t *bal_colP,
                               const int32_t *bal_colQ, const double *bal_coef,
                               int64_t nnzj, const int64_t *jrow, const int64_t *jcol,
                               int64_t nnzh, const int64_t *hrow, const int64_t *hcol,
                               const double *xL, const double *xU, const double *gL,
                               const double *gU, int nsh, const int32_t *sh_bus, const double *sh_gs,
                               const double *sh_bs, int ndc, const double *dc_loss1);
/* ... and in the W-space form of examples/acopf/acwr.jl (acopf_synth.py, acwr_layout) */
ora_problem *ora_problem_acopf_acwr(int nb, int ng, int nl, const int32_t *f_bus,
                               const int32_t *t_bus, const double *ohm /* [nl][12] */,
                               const int32_t *gen_bus, const double *c2,
                               const double *c1, const int32_t *bal_ptr, const int32_t *bal_colP,
                               const int32_t *bal_colQ, const double *bal_coef,
                               int64_t nnzj, const int64_t *jrow, const int64_t *jcol,
                               int64_t nnzh, const int64_t *hrow, const int64_t *hcol,
                               const double *xL, const double *xU, const double *gL,
                               const double *gU, int nsh, const int32_t *sh_bus, const double *sh_gs,
                               const double *sh_bs, int ndc, const double *dc_loss1,
                                    int nbp, const int32_t *bp_i, const int32_t *bp_j, const int32_t *br_bp,
                                    const double *br_sig, const double *bp_tmin, const double *bp_tmax);
const ora_nlp *ora_problem_nlp(const ora_problem *p);
const double *ora_problem_x0(const ora_problem *p);
void ora_problem_destroy(ora_problem *p);

#ifdef __cplusplus
}
#endif
#endif
