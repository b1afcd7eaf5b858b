/*
 * sqphip.h -- C ABI of libsqphip.so, the MI355X (gfx950) hot path behind SqpSolver.Optimizer.
 *
 * What it replaces in /root/reference (exanauts/SqpSolver.jl):
 *   - the per-iteration QP sub-problem that `SqpTR` delegates to `external_optimizer` through
 *     `QpJuMP` (src/algorithms/subproblem_JuMP.jl:127-183 QP, :185-244 LP phase, :283-347 L1QP,
 *     :352-393 feasibility restoration, :398-429 INFEAS, :432-463 trust-region bounds,
 *     :514-563 collect_solution!), i.e. the `AbstractSubOptimizer` seat of
 *     src/algorithms/subproblem.jl:1-28 as dispatched by src/algorithms/sqp_trust_region.jl:314-331;
 *   - the merit / step-acceptance path of src/algorithms: common.jl:14-77, merit.jl:13-17,
 *     sqp.jl:170-213, sqp_trust_region.jl:370-380 and :487-579.
 * The Julia host (MOI_wrapper.jl, model.jl) stays; it reaches this library through `ccall`
 * (binding shown in INTEGRATION.md).  Plain pointers and sizes only; no C++ or torch types.
 *
 * Conventions
 *   - every function returns 0 on success, a negative SQPHIP_E* code on misuse / HIP failure;
 *     nothing throws across the ABI; sqphip_last_error() gives the message.
 *   - solver outcomes are MOI.TerminationStatusCode integers (sub-problem) and the Ipopt-style
 *     codes of src/status.jl:2-23 (whole solve).
 *   - all arrays are caller-owned HOST memory unless the name ends in _dev; indices are 1-based
 *     (Julia-native) at the boundary.
 *   - multipliers use the JuMP sign convention exactly as collect_solution! returns them
 *     (mult_x_U <= 0 <= mult_x_L; stationarity H p + c = J'lambda + mult_x_L + mult_x_U).
 *   - one HIP stream per context; a context is not thread-safe, distinct contexts are.
 */
#ifndef SQPHIP_H
#define SQPHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sqphip_ctx sqphip_ctx;

enum { SQPHIP_OK = 0, SQPHIP_EINVAL = -1, SQPHIP_EHIP = -2, SQPHIP_ENOMEM = -3, SQPHIP_ESTATE = -4 };

/* sub-problem modes (subproblem_JuMP.jl line ranges above) */
enum { SQPHIP_MODE_QP = 0, SQPHIP_MODE_FR = 1, SQPHIP_MODE_SOC = 2, SQPHIP_MODE_LP = 3,
       SQPHIP_MODE_L1QP = 4, SQPHIP_MODE_INFEAS = 5 };

/* MOI.TerminationStatusCode values this library can return (MathOptInterface v1 enum order) */
enum { SQPHIP_MOI_LOCALLY_SOLVED = 4, SQPHIP_MOI_LOCALLY_INFEASIBLE = 5,
       SQPHIP_MOI_ITERATION_LIMIT = 11, SQPHIP_MOI_NUMERICAL_ERROR = 20 };

/* Mirror of src/parameters.jl:1-30 (fields the hot path reads) plus the sub-solver's own knobs */
typedef struct {
    double tol_direction, tol_residual, tol_infeas;   /* parameters.jl:17-19 */
    double init_mu, max_mu, tr_size;                   /* :22,:23,:28 */
    double rho, eta, tau, min_alpha;                   /* :24-27 (Armijo variant) */
    int32_t max_iter;                                  /* :20 */
    int32_t use_soc;                                   /* :29 */
    int32_t literal_quirks;  /* 1: reproduce SURVEY.md App. C #2/#3 (JuMP-sign Hessian/KT residual) */
    double ipm_tol;          /* interior-point optimality tolerance (scaled), default 1e-9 */
    int32_t ipm_max_iter;    /* default 200; a second-order correction (mode 2) gets half of it: a correction that has not
                              * converged by then is abandoned -- for run! that is the same as any other unsuccessful
                              * correction (no correction step, sqp_trust_region.jl:341-360), and no successful one of a
                              * 5082-sub-problem survey of the IEEE-118 workload needed more than 83 iterations */
    int32_t ipm_phase1;      /* 1: confirm infeasibility verdicts with a phase-1 run (default 0) */
    int32_t device;          /* HIP device ordinal */
    int32_t ipm_corrector;   /* 0 (default since round 4): monotone Fiacco-McCormick barrier rule throughout -- what Ipopt,
                              * the sub-solver of every test and example of the reference, does by default (mu_strategy =
                              * monotone; test/ext_solver.jl:2-6 and examples/acopf/opf.jl:59-64 leave it there): one solve
                              * per factorisation; 1: Mehrotra predictor-corrector iterations until the first inertia
                              * correction of a solve (4 % fewer factorisations on the IEEE-118 workload, a second solve per
                              * iteration: 15 - 25 % slower end to end) */
    int32_t kkt_condense;    /* 1: rows with gL != gU (diagonal block -D of the Newton matrix) are eliminated before
                              * the factorisation: dense LDL^T of order n + #(gL == gU) instead of n + m */
    int32_t kkt_tile_order;  /* 1 (needs kkt_condense): the variables are ordered so that the leading tile columns of
                              * the condensed matrix are mutually independent (sqphip_kkt_order, rows_last = 1) and
                              * the factorisation treats them as such: one launch per kernel for all of them, the
                              * dense chain only on the remainder */
    int32_t kkt_mode;        /* linear solver of the Newton systems.  1: batched dense LDL^T on the MFMA pipe (ldlt.hip);
                              * 2: multifrontal LDL^T of the sparse matrix (mfront.hip: symbolic analysis once per
                              * structure, dense fronts in LDS); 0 (default): the sparse one when it does at most a
                              * quarter of the dense flops and no front exceeds 512 rows, else the dense one */
    int32_t ipm_warm_start;  /* 1: the first interior-point run of a sub-problem starts from the step and the equality-row
                              * multipliers of the instance's previous solved sub-problem of the same mode -- what
                              * warm_start_init_point = "yes" asks of Ipopt (/root/reference/test/ext_solver.jl:5,
                              * examples/acopf/opf.jl:62; upstream the model is rebuilt every iteration, so the option has
                              * no effect there).  0 (default): centred cold start, fewer iterations on every workload
                              * measured */
} sqphip_options;

void sqphip_default_options(sqphip_options *o);

/* Create a context for `batch` NLP instances sharing dimensions and sparsity
 * (one instance = one SqpSolver.Model, src/model.jl:37-67).  COO structures as produced by
 * MOI_wrapper.jl:930-945 (Jacobian) and :1010-1025 (Hessian, triangular, duplicates allowed);
 * nnzH = 0 means no Hessian (LP / SLP).  Bounds may be +-Inf; they apply to every instance
 * until overridden with sqphip_set_bounds.  With options.kkt_condense = 1 the rows with gL == gU given HERE are the
 * ones that stay in the factorised matrix: sqphip_set_bounds may change their values per instance but returns
 * SQPHIP_EINVAL if it would turn one of the other rows into an equality. */
int sqphip_create(sqphip_ctx **ctx, int64_t n, int64_t m, int64_t num_linear,
                  int64_t nnzJ, const int64_t *jrow, const int64_t *jcol,
                  int64_t nnzH, const int64_t *hrow, const int64_t *hcol,
                  const double *xL, const double *xU, const double *gL, const double *gU,
                  const sqphip_options *opt, int32_t batch);
void sqphip_destroy(sqphip_ctx *ctx);
/* Host-only (no GPU): the ordering options.kkt_tile_order = 1 gives the condensed Newton matrix of this structure.
 * pos[u], u < n + #(gL == gU): position of variable u (u < n) or of the (u - n)-th row with gL == gU in the
 * factorised matrix; the first n_lead_tiles 64-column tiles are mutually independent (block-diagonal leading block,
 * identity padding inside the tiles), the positions from 64 * n_lead_tiles to order - 1 are the dense remainder.
 * rows_last = 1 is what the library uses: rows enter a leading tile only behind ALL the variables they couple to,
 * everything else goes to the remainder.  rows_last = 0 lets rows and variables compete freely for the tiles
 * (smaller remainder, but rows pivoted before their variables lose digits -- kept for experiments only).
 * Same COO conventions as sqphip_create. */
int sqphip_kkt_order(int64_t n, int64_t m, int64_t nnzJ, const int64_t *jrow, const int64_t *jcol,
                     int64_t nnzH, const int64_t *hrow, const int64_t *hcol, const double *gL, const double *gU,
                     int32_t rows_last, int32_t *pos, int32_t *n_lead_tiles, int32_t *order);
/* Host-only (no GPU): symbolic analysis of the sparse Newton matrix of this structure -- what the analysis phase of
 * Ipopt's linear solver does for the reference (/root/reference/examples/acopf/opf.jl:59-64).  condense = 1: rows with
 * gL != gU (and at most 32 entries) are eliminated first, as options.kkt_condense does.  rows_after_vars = 1 is what
 * the library uses: a row is ordered behind every variable it couples to.  small_front / zero_frac: amalgamation
 * thresholds (<= 0 / < 0: library defaults).  pos[u], u < order: position of variable u (u < n) or of the (u - n)-th
 * kept row in the elimination order; any output may be NULL. */
typedef struct {
    int64_t order;            /* unknowns of the factorised matrix */
    int64_t nnz_k_lower;      /* structural entries of its lower triangle, diagonal included */
    int64_t n_supernodes, n_levels, max_front, max_cols;
    int64_t nnz_l;            /* entries of L below the diagonal as the dense fronts hold them (explicit zeros included) */
    int64_t nnz_l_exact;      /* ... of the exact sparse factor under the same order */
    double flops, flops_exact;/* 2 x multiply-adds of one numeric factorisation: dense fronts / exact sparse */
    int64_t front_doubles;    /* doubles of front storage per instance */
} sqphip_symbolic_stats;
int sqphip_kkt_symbolic(int64_t n, int64_t m, int64_t nnzJ, const int64_t *jrow, const int64_t *jcol,
                        int64_t nnzH, const int64_t *hrow, const int64_t *hcol, const double *gL, const double *gU,
                        int32_t condense, int32_t rows_after_vars, int32_t small_front, double zero_frac,
                        int32_t *pos, sqphip_symbolic_stats *out);
const char *sqphip_last_error(const sqphip_ctx *ctx);
int sqphip_set_bounds(sqphip_ctx *ctx, int32_t inst, const double *xL, const double *xU,
                      const double *gL, const double *gU);

/* ---- the AbstractSubOptimizer seat ------------------------------------------------------------
 * Replaces sub_optimize! / sub_optimize_FR! / sub_optimize_lp / sub_optimize_L1QP! /
 * sub_optimize_infeas and sub_optimize_soc! (pass mode SOC with E = E_soc,
 * sqp_trust_region.jl:341-360).  Jval/Hval are in the COO order given to sqphip_create
 * (what eval_jac_g / eval_h fill, sqp.jl:93,112); Hval may be NULL.
 * Outputs: p[n], lambda[m], mult_x_U[n], mult_x_L[n] caller-allocated; slack may be NULL else [2m]
 * (t+ then t-).  For mode LP `p` receives the absolute point x (subproblem_JuMP.jl:212-218).
 * Infeasible-family statuses zero the outputs (subproblem_JuMP.jl:551-555). */
int sqphip_qp_solve(sqphip_ctx *ctx, int32_t mode, const double *x_k, double delta, double mu,
                    const double *df, const double *E, const double *Jval, const double *Hval,
                    double *p, double *lambda, double *mult_x_U, double *mult_x_L,
                    double *slack, int32_t *moi_status);
/* statistics of the last sqphip_qp_solve: interior-point iterations, KKT factorisations */
int sqphip_qp_stats(const sqphip_ctx *ctx, int32_t *ipm_iters, int32_t *n_factor);
/* ... how its last interior-point run ended: rule 0 = scaled optimality error <= options.ipm_tol, 1 / 2 / 3 = one of the
 * acceptable-termination rules (8 consecutive iterates within 100 x the tolerance, 15 within 1 000 x, 25 within 10^4 x:
 * what Ipopt's acceptable_tol / acceptable_iter do for the reference's sub-solver, test/ext_solver.jl:2-6), -1 = not
 * converged; scaled_error = that error at the final iterate.  A sub-problem reported LOCALLY_SOLVED under rule 3 is
 * accurate to 1e4 x ipm_tol only. */
int sqphip_qp_termination(const sqphip_ctx *ctx, int32_t *rule, double *scaled_error);

/* ---- merit / acceptance path (device reductions over host-supplied vectors) -------------------- */
/* common.jl:54-77; pnorm 1, 2 or 0 (=Inf) */
int sqphip_norm_violations(sqphip_ctx *ctx, const double *E, const double *x, int32_t pnorm,
                           double *out);
/* common.jl:14-23 */
int sqphip_kt_residuals(sqphip_ctx *ctx, const double *df, const double *lambda,
                        const double *mult_x_U, const double *mult_x_L, const double *Jval,
                        double *out);
/* common.jl:30-47 */
int sqphip_norm_complementarity(sqphip_ctx *ctx, const double *E, const double *lambda,
                                int32_t pnorm, double *out);
/* sqp.jl:170-183 given f(x+ap), g(x+ap) from the host callbacks */
int sqphip_compute_phi(sqphip_ctx *ctx, double f_trial, const double *E_trial,
                       const double *x_trial, double mu, int32_t feasibility_restoration,
                       double *phi);
/* sqp_trust_region.jl:487-508: q(p) if with_step else q(0) */
int sqphip_compute_qmodel(sqphip_ctx *ctx, const double *x, const double *p, const double *df,
                          const double *E, const double *Jval, const double *Hval, double mu,
                          int32_t with_step, double *q);
/* merit.jl:13-17 + sqp.jl:190-213 (scalar-mu form): D = df'p - mu * sum(viol(E)) */
int sqphip_compute_derivative(sqphip_ctx *ctx, const double *df, const double *p,
                              const double *E, double mu, double *D);
/* The whole compute_derivative(sqp) of sqp.jl:190-213 over merit.jl:13-17.  mu_vec (may be NULL): the vector-penalty
 * forms D = dfp - mu_vec' cons_viol (merit.jl:14,17), else D = dfp - mu sum(cons_viol) (:15,:16).
 * feasibility_restoration = 1 (sqp.jl:194-203): dfp = sum of the slack values of the last sub-problem (slack[2m] as
 * sqphip_qp_solve returns them), cons_viol_i = violation of E_i - viol_i. */
int sqphip_compute_derivative_full(sqphip_ctx *ctx, const double *df, const double *p, const double *E, double mu,
                                   const double *mu_vec, int32_t feasibility_restoration, const double *slack,
                                   double *D);
/* sqp_trust_region.jl:529-538,:574-577: ratio test and radius update.
 * accept_out = 1 if ared > 0 and ared/pred > 0; delta_out the updated radius. */
int sqphip_tr_update(double ared, double pred, double delta, double pnorm_inf,
                     double delta_max, double tol_direction, int32_t *accept_out,
                     double *delta_out);

/* sqp_line_search.jl:303-334 (compute_alpha; the line-search algorithm is unreachable upstream, kept for
 * completeness of the merit path): backtracking Armijo search alpha <- tau * alpha while
 * phi(alpha) > phi0 + eta * alpha * D, starting from 1; stops with is_valid = 0 once alpha < min_alpha.
 * phi(alpha) is the caller's merit at x + alpha p (compute_phi needs eval_f / eval_g: host callbacks), so it
 * comes in as a C callback.  Returns at once with alpha = 1, is_valid = 1 when pnorm_inf <= tol_direction. */
typedef double (*sqphip_phi_fn)(double alpha, void *user);
int sqphip_armijo_alpha(double phi0, double D, double eta, double tau, double min_alpha, double pnorm_inf,
                        double tol_direction, sqphip_phi_fn phi, void *user, double *alpha, int32_t *is_valid);
/* sqp_line_search.jl:270-294 (compute_mu_rule1! / rule2! / rule3!), vector penalty mu[m]:
 *   t = (df'p + max(p'Hp/2, 0)) / max((1 - rho) * viol1, 1e-8)
 *   rule 1: mu_i = max(mu_i, t, |lambda_i|);  rule 2: iter == 1 ? mu_i = t : mu_i = max(mu_i, |lambda_i|);
 *   rule 3: mu_i = max(mu_i, |lambda_i|) */
int sqphip_compute_mu_rule(int32_t rule, int64_t iter, double rho, double viol1, double dfp, double half_pHp,
                           int64_t m, const double *lambda, double *mu);
/* The same rules with every reduction on the device (wave-level butterflies + LDS exchange): norm_violations(sqp, 1),
 * df'p and p'Hp/2 from the iterate (x, E, df, p, Hval in the COO order of the structure; Hval may be NULL), then the
 * element-wise update of mu[m] (in/out, host memory). */
int sqphip_compute_mu_rule_dev(sqphip_ctx *ctx, int32_t rule, int64_t iter, double rho, const double *x, const double *E,
                               const double *df, const double *p, const double *Hval, const double *lambda, double *mu);
/* compute_alpha (sqp_line_search.jl:303-334) for an instance of the built-in ACOPF evaluator, entirely on the device:
 * phi(alpha) = f(x + alpha p) + mu |viol(x + alpha p)|_1 (|viol|_1 alone under feasibility restoration,
 * sqp.jl:170-183) is evaluated by the device callbacks, the backtracking loop alpha <- tau alpha runs inside the kernel
 * and stops with is_valid = 0 once alpha < min_alpha.  n_eval = merit evaluations made. */
int sqphip_acopf_armijo(sqphip_ctx *ctx, int32_t inst, const double *x, const double *p, double mu, double phi0,
                        double D, double eta, double tau, double min_alpha, int32_t feasibility_restoration,
                        double *alpha, int32_t *is_valid, int32_t *n_eval);

/* ---- device-resident batched SQP-TR over the built-in ACOPF evaluator --------------------------
 * (sqp_trust_region.jl:98-223 for every instance of the batch; callbacks of
 * MOI_wrapper.jl:1115-1146 replaced by HIP kernels over PowerModels-ACP-shaped data,
 * test/opf.jl:5-9).  Topology is shared; electrical data, loads and bounds are per instance. */
int sqphip_acopf_attach(sqphip_ctx *ctx, int32_t nb, int32_t ng, int32_t nl,
                        const int32_t *f_bus, const int32_t *t_bus, const int32_t *gen_bus,
                        const int32_t *bal_ptr, const int32_t *bal_colP, const int32_t *bal_colQ,
                        const double *bal_coef, int32_t ref_bus);
/* The same evaluator in rectangular voltage coordinates: PowerModels' ACRPowerModel under the build_opf of
 * /root/reference/examples/acopf/opf.jl:12-43 -- the formulation run_sqp_opf instantiates (:46, :51).  Variables
 * (vi, vr, pg, qg, flows, dc lines), rows vi[ref] = 0; power balance; vmin^2 <= vr^2 + vi^2 and vr^2 + vi^2 <= vmax^2 per
 * bus (constraint_voltage_magnitude_bounds); thermal limits; Ohm's law with v_f v_t cos / sin written as
 * vr_f vr_t + vi_f vi_t and vi_f vr_t - vr_f vi_t (same twelve coefficients per branch); dc-line losses; no
 * angle-difference rows (opf.jl:33).  The context must have been created with the structure of
 * sqpsolver.jl_amd/acopf_synth.py acr_layout; everything after the attach (set_shunts: four Jacobian and two Hessian
 * entries per shunted bus, num_linear = 1; set_dclines; set_instance; eval; sqp_run) is shared with the polar form. */
int sqphip_acopf_attach_acr(sqphip_ctx *ctx, int32_t nb, int32_t ng, int32_t nl,
                            const int32_t *f_bus, const int32_t *t_bus, const int32_t *gen_bus,
                            const int32_t *bal_ptr, const int32_t *bal_colP, const int32_t *bal_colQ,
                            const double *bal_coef, int32_t ref_bus);
/* ... and in the W-space form of /root/reference/examples/acopf/acwr.jl:1-37 (ACWRPowerModel over PowerModels' build_opf;
 * defined by the reference, instantiated by none of its scripts): variables (vi, vr, w, wr, wi, pg, qg, flows, dc lines)
 * with w_i = |v_i|^2 and wr, wi per bus pair i < j; balance, angle-difference (wi <= tan(angmax) wr, wi >= tan(angmin) wr)
 * and Ohm rows linear in (w, wr, wi); constraint_model_voltage as nb + 2 nbp quadratic equalities; thermal limits.
 * Structure: sqpsolver.jl_amd/acopf_synth.py acwr_layout.  bp_i / bp_j: the buses of pair k; br_bp / br_sig: pair and
 * orientation (+1: from = i, -1: from = j) of branch l; bp_tmin / bp_tmax: tan of the pair's angle limits.  Shunts enter
 * the (linear) balance rows through w: sqphip_acopf_set_shunts supplies them without changing the structure. */
int sqphip_acopf_attach_acwr(sqphip_ctx *ctx, int32_t nb, int32_t ng, int32_t nl,
                             const int32_t *f_bus, const int32_t *t_bus, const int32_t *gen_bus,
                             const int32_t *bal_ptr, const int32_t *bal_colP, const int32_t *bal_colQ,
                             const double *bal_coef, int32_t ref_bus, int32_t nbp, const int32_t *bp_i,
                             const int32_t *bp_j, const int32_t *br_bp, const double *br_sig, const double *bp_tmin,
                             const double *bp_tmax);
/* Bus shunts (optional, after sqphip_acopf_attach): bus sh_bus[s] consumes gs[s] vm^2 of active and injects
 * bs[s] vm^2 of reactive power.  The context must have been created with the matching structure: two more Jacobian
 * COO entries (P row, Q row; column vm) and one more Hessian COO entry (vm, vm) per shunted bus at the END of the
 * lists, and num_linear = 2 nl + 1 (the balance rows are no longer linear) -- sqpsolver.jl_amd/acopf_synth.py,
 * acopf_layout.  Shared by every instance of the batch. */
int sqphip_acopf_set_shunts(sqphip_ctx *ctx, int32_t nsh, const int32_t *sh_bus, const double *gs, const double *bs);
/* HVDC lines (optional, after sqphip_acopf_attach; PowerModels variable_dcline_power +
 * constraint_dcline_power_losses, /root/reference/examples/acopf/opf.jl:16,40-42).  The structure given to
 * sqphip_create carries them already -- per line 4 variables (p_f, p_t, q_f, q_t of the line, blocks of ndc behind
 * all other variables, entering the balance rows of their buses through the incidence lists of sqphip_acopf_attach)
 * and one row  (1 - loss1) p_f + p_t = loss0  behind all other rows, its two Jacobian entries behind all others
 * (acopf_synth.py, acopf_layout); this call supplies loss1 per line (default 0).  ndc must match the structure. */
int sqphip_acopf_set_dclines(sqphip_ctx *ctx, int32_t ndc, const double *loss1);
/* ohm[nl][12]: per branch the coefficients (A, Bc, Bs) of the four flow equations p_f, q_f, p_t, q_t,
 *   F_k = A_k v_self^2 + v_f v_t (Bc_k cos(va_f - va_t) + Bs_k sin(va_f - va_t)),
 * i.e. the pi model with an ideal transformer (tap ratio, phase shift) at the from end folded into 12 numbers on the
 * host (sqpsolver.jl_amd/acopf_synth.py, Network.branch_coeffs); an outaged branch is 12 zeros. */
int sqphip_acopf_set_instance(sqphip_ctx *ctx, int32_t inst, const double *ohm, const double *c2, const double *c1,
                              const double *x0);
/* Evaluate the five callbacks on the device for instance `inst` at host point x (parity tests).
 * Any output may be NULL. lambda/sigma only matter for hval. */
int sqphip_acopf_eval(sqphip_ctx *ctx, int32_t inst, const double *x, double sigma,
                      const double *lambda, double *f, double *grad, double *g, double *jval,
                      double *hval);
/* A synthetic NLP with a DENSE Lagrangian Hessian for the batched run (bench.py --workload dense; no reference counterpart:
 * the reference's examples are ACOPF models, whose Hessians are sparse):
 *     min  1/2 x'Qx + c'x + kappa / 4 sum_i x_i^4    s.t.  A x = b,  xL <= x <= xU
 * Q [n][n] symmetric and A [m][n] row-major are shared by the batch; an instance carries c [n], its start x0 and -- through
 * sqphip_set_bounds -- xL, xU and gL = gU = b.  The context must have been created with the matching structure
 * (sqpsolver.jl_amd/dense_synth.py): Jacobian COO = A row-major (m n entries), Hessian COO = lower triangle of Q
 * column-major (n (n + 1) / 2 entries), num_linear = m.  With options.kkt_mode = 1 every sub-problem factorises a dense
 * Newton matrix of order n + m on the MFMA path (ldlt.hip). */
int sqphip_dense_attach(sqphip_ctx *ctx, const double *Q, const double *A, double kappa);
int sqphip_dense_set_instance(sqphip_ctx *ctx, int32_t inst, const double *c, const double *x0);
/* Run SQP-TR for every instance until each has terminated or done `max_outer` more outer
 * iterations (0 = no cap beyond options.max_iter).  Restartable: state stays on the device. */
int sqphip_sqp_reset(sqphip_ctx *ctx);
int sqphip_sqp_run(sqphip_ctx *ctx, int32_t max_outer);
/* results per instance (src/model.jl result slots as written by sqp_trust_region.jl:215-222);
 * any pointer may be NULL */
/* Scenario queue: more scenarios than the context has slots (contingency screening).  A slot whose run has terminated
 * files its result under its scenario id, takes the next id from a device-wide counter, loads that scenario from tables
 * in HBM and starts over -- inside the kernels of the running sweep, without the host -- so the batch stays full until
 * the queue is empty and no slot waits for the slowest run of a batch (no reference counterpart; SURVEY.md section
 * 8f-4, load re-balancing of stragglers).  Which slot solves which scenario depends on timing, the result of a
 * scenario does not.  _begin allocates tables for n_scenarios; _set uploads one scenario (the arguments of
 * sqphip_set_bounds and sqphip_acopf_set_instance); _run solves them all; _get returns one result (final point,
 * objective, status as src/status.jl, iterations; iterations = -1: this context has filed no result for the scenario since
 * the last _stream_run / _stream_assign -- another rank solved it, or it is still waiting or in progress). */
int sqphip_sqp_stream_begin(sqphip_ctx *ctx, int32_t n_scenarios);
int sqphip_sqp_stream_set(sqphip_ctx *ctx, int32_t scenario, const double *xL, const double *xU, const double *gL,
                          const double *gU, const double *ohm, const double *c2, const double *c1, const double *x0);
int sqphip_sqp_stream_run(sqphip_ctx *ctx);
/* A queue shared between ranks (multi-GPU screening): every rank uploads the tables of ALL
 * scenarios (_begin with the total, _set for each) but hands out only the ids assigned to it (_assign: ids in hand-out
 * order).  _run_some lets every slot perform up to max_outer more outer iterations and reports how many ids of this
 * rank's queue nobody has drawn yet and how many slots have a run in progress; between such runs the host may take
 * unstarted ids off the tail of a rank's queue (_release) and hand them to a rank whose queue ran dry (_append) --
 * sqpsolver.jl_amd/shard.py run_shared_queue does that with one small host-side exchange per round.  No iterate crosses
 * ranks, and the result of a scenario does not depend on the rank or slot that solved it. */
int sqphip_sqp_stream_assign(sqphip_ctx *ctx, int32_t n, const int32_t *ids);
int sqphip_sqp_stream_append(sqphip_ctx *ctx, int32_t n, const int32_t *ids);
int sqphip_sqp_stream_release(sqphip_ctx *ctx, int32_t n, int32_t *ids_out, int32_t *n_out);
int sqphip_sqp_stream_run_some(sqphip_ctx *ctx, int32_t max_outer, int32_t *n_unstarted, int32_t *n_active);
int sqphip_sqp_stream_get(sqphip_ctx *ctx, int32_t scenario, double *x, double *obj_val, int32_t *status, int32_t *iter);
int sqphip_sqp_get(sqphip_ctx *ctx, int32_t inst, double *x, double *g, double *mult_g,
                   double *mult_x_L, double *mult_x_U, double *obj_val, int32_t *status,
                   int32_t *iter);
/* ret codes and iteration counts of the whole batch, device -> host (the arrays a host layer
 * all-gathers across ranks); done[i] = 1 once instance i has terminated */
int sqphip_sqp_status(sqphip_ctx *ctx, int32_t *ret_codes, int32_t *iters, int32_t *done);
/* ---- multi-GPU: the convergence-status gather --------------------------------------
 * Independent instances are cut into `world` contiguous blocks (sizes differing by at most one, rank r holding block r;
 * one process and one context per GPU, the context's batch = its block).  The only exchange between ranks is an
 * all-gather of int32 (ret, iter, done) per instance over RCCL on a communicator the library owns: rank 0 obtains the
 * 128-byte ncclUniqueId with sqphip_comm_unique_id and ships it to the other ranks by whatever channel the host has
 * (MPI, a file, torch.distributed, Julia's Distributed), every rank then calls sqphip_comm_init.  RCCL is loaded with
 * dlopen on the first of these calls; a context without a communicator returns the local table from
 * sqphip_gather_status, one with a communicator always runs the collective (also for world = 1).  `total` must be the
 * same on every rank of a call (the all-gather counts derive from it).  sqphip_comm_init is collective: call
 * sqphip_comm_available (1 = librccl loads in this process) on every rank and agree on the minimum before entering it.
 * Outputs: length `total`, ordered by global instance id; any may be NULL. */
int sqphip_comm_available(void);
int sqphip_comm_unique_id(void *id128);
int sqphip_comm_init(sqphip_ctx *ctx, const void *id128, int32_t world, int32_t rank);
int sqphip_gather_status(sqphip_ctx *ctx, int32_t total, int32_t *ret_codes, int32_t *iters, int32_t *done);
int sqphip_comm_destroy(sqphip_ctx *ctx);
/* per-instance trace rows (columns of the reference's log line, sqp_trust_region.jl:605-634):
 * rows[k*12 + {iter, accepted, fr, sub_status, ipm_iters, f, phi, mu, delta, |p|, inf_pr, inf_du}] */
int sqphip_sqp_trace(sqphip_ctx *ctx, int32_t inst, double *rows, int32_t cap, int32_t *len);

/* counters since create/reset: sub-problem solves, interior-point iterations, factorisations,
 * flops spent in LDL^T (N^3/3 each), seconds inside the factor kernels (HIP events) */
typedef struct {
    int64_t n_qp, n_ipm_iter, n_factor;
    double ldlt_flops, ldlt_seconds, trailing_seconds, solve_seconds, total_seconds;
    int64_t trailing_launches;
    int64_t kkt_order;       /* order of the matrices the LDL^T factorises (n + m, or the condensed order) */
    int64_t lead_tiles;      /* leading 64-column tiles treated as mutually independent (options.kkt_tile_order) */
    double trailing_flops_per_factor;   /* algorithmic flops of the k_trailing launches of one factorisation */
    /* sparse (multifrontal) solver; all zero when the dense one is in use */
    int64_t sparse;          /* 1: the Newton systems go through mfront.hip */
    int64_t nnz_k;           /* structural entries of the lower triangle of the factorised matrix, diagonal included */
    int64_t nnz_l;           /* entries of L below the diagonal as the fronts hold them (deterministic order of the library) */
    int64_t n_supernodes, n_levels, max_front;
    double factor_flops;     /* flops of one numeric factorisation of one instance (dense partial factorisations of the fronts) */
    int64_t front_doubles;   /* doubles of front storage per instance (L + contribution blocks + right-hand-side rows) */
    int64_t cb_doubles;      /* ... of which contribution blocks (written once by the child, read once by the parent) */
    int64_t factor_launches, solve_launches;   /* kernel launches of one factorisation / of one forward + backward solve */
    int64_t n_sweeps;        /* passes of the fixed kernel sequence (ipm_sweep) since create / reset: with continuous
                              * batching the slowest instance of the batch decides this number */
    int64_t n_solve;         /* forward + backward solves with the factors, summed over the instances (first solve of a
                              * factorisation, corrector, refinement steps) */
    int64_t n_groups;        /* instance groups of sqphip_sqp_run, each on its own HIP stream and host thread (1: none).
                              * n_sweeps and the kernel seconds are summed over the groups */
    int64_t nnz_l_top, nnz_k_top, cols_top;   /* the narrow top of the assembly tree (the fronts k_mf_solve_top2 streams): entries
                              * of L, structural entries of the matrix, columns -- the algorithmic bytes of the per-kernel records */
} sqphip_counters;
int sqphip_get_counters(sqphip_ctx *ctx, sqphip_counters *c);
/* The work of the batched run since sqphip_sqp_reset, split by sub-problem mode: out[3 k + 0..2] = sub-problems solved,
 * interior-point iterations, KKT factorisations of mode k = 0 QP (sub_optimize!, subproblem_JuMP.jl:127-183), 1 FR
 * (:352-393), 2 SOC (sqp_trust_region.jl:341-360), 3 linear phase (:264-304).  12 values. */
int sqphip_get_mode_counters(sqphip_ctx *ctx, int64_t *out12);
/* ... and per instance (arrays of length batch, any may be NULL): sub-problems, interior-point iterations and KKT
 * factorisations since sqphip_sqp_reset.  With continuous batching a call to sqphip_sqp_run lasts as long as its
 * busiest instance: max / mean of `factorisations` is the load imbalance of the batch. */
int sqphip_sqp_work(sqphip_ctx *ctx, int64_t *sub_problems, int64_t *ipm_iterations, int64_t *factorisations);
/* The last (up to 64) sub-problems of one instance in the order they finished: rows of four int32 (mode, MOI status,
 * interior-point iterations, factorisations); *n_rows receives the number written (cap: rows available in `rows`). */
int sqphip_sqp_qp_log(sqphip_ctx *ctx, int32_t inst, int32_t *rows, int32_t cap, int32_t *n_rows);
/* ... and for the same rows: the scaled optimality error each ended with and the rule that ended it (as
 * sqphip_qp_termination; either array may be NULL) */
int sqphip_sqp_qp_log_term(sqphip_ctx *ctx, int32_t inst, double *scaled_error, int32_t *rule, int32_t cap, int32_t *n_rows);
/* Sub-problems of the batched run since sqphip_sqp_reset by the way their interior-point run ended: out4[0] tolerance
 * reached, out4[1..3] acceptable-termination rule 1 / 2 / 3 (sqphip_qp_termination). */
int sqphip_get_termination_counters(sqphip_ctx *ctx, int64_t *out4);
/* Diagnostics: the sub-problem request instance `inst` of the batched run worked on last, in the argument convention of
 * sqphip_qp_solve (x_k[n], c[n], b[m], jac_coo[nnzJ], hess_coo[nnzH]; any pointer may be NULL), so that a sub-problem
 * seen on the device can be replayed through sqphip_qp_solve or another solver. */
int sqphip_sqp_last_request(sqphip_ctx *ctx, int32_t inst, int32_t *mode, double *delta, double *mu_pen, double *x_k,
                            double *c, double *b, double *jac_coo, double *hess_coo);
int sqphip_reset_counters(sqphip_ctx *ctx);
/* HIP-event timing of the factor / trailing-update / solve kernels (off by default); enabled = 2 additionally brackets every
 * launch group of a sweep by kernel class (about twenty more event records per sweep: meant for a short measurement leg) */
int sqphip_set_timing(sqphip_ctx *ctx, int32_t enabled);
/* ... the class timers (sparse path): seconds[c] of kernel time and groups[c] = launch groups timed, summed over the instance
 * groups, since sqphip_reset_counters; classes c = 0 values of the matrix entries (k_mf_values), 1 front kernels below the
 * narrow top of the assembly tree, 2 front kernels of the top (level launches, or k_mf_spine), 3 the top of the tree in
 * the solves (k_mf_solve_top2, with the inertia test), 4 level launches of the solves (k_mf_fwd2 / k_mf_bwd2), 5 the stage
 * kernel behind the solve (k_ipm_post: residual check, step, convergence test, next right-hand side), 6 transitions between
 * sub-problems (k_qp_finish, k_sqp_stage, k_ipm_head).  cap = length of both arrays (7 classes). */
int sqphip_get_kernel_times(sqphip_ctx *ctx, double *seconds, int64_t *groups, int32_t cap);

/* Test hooks and micro-benchmarks (host reference of the multifrontal plan, kernel-level twins, dense LDL^T probes) are
 * exported by the library but are not part of the drop-in boundary: include/sqphip_test_hooks.h. */

#ifdef __cplusplus
}
#endif
#endif
