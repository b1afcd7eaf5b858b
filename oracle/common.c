/* common.c -- oracle restatement of the scalar reductions.  TEST INFRASTRUCTURE ONLY.
 * Follows /root/reference/src/algorithms/common.jl and merit.jl. */
#include "sqp_oracle.h"
#include <math.h>
#include <stdlib.h>

void ora_default_options(ora_options *o)
{
    /* src/parameters.jl:17-29 */
    o->tol_direction = 1e-8;
    o->tol_residual = 1e-8;
    o->tol_infeas = 1e-8;
    o->max_iter = 3000;
    o->init_mu = 1.0;
    o->max_mu = 1e10;
    o->rho = 0.8;
    o->eta = 0.4;
    o->tau = 0.9;
    o->min_alpha = 1e-6;
    o->tr_size = 10.0;
    o->use_soc = 0;
    o->literal_quirks = 1;
    o->ipm_tol = 1e-9;
    o->ipm_max_iter = 200;
    o->ipm_phase1 = 0;
    o->num_threads = 1;
    o->ipm_corrector = 0;   /* monotone rule: Ipopt's default mu_strategy (the reference leaves it there) */
    o->kkt_condense = 1;
    o->kkt_tile_order = 0;
    o->kkt_mode = 0;
    o->ipm_warm_start = 0;
}

/* Julia's isapprox(a, b) with default rtol = sqrt(eps), atol = 0
 * (used at sqp_trust_region.jl:146, :200, :535) */
int ora_isapprox(double a, double b)
{
    if (a == b) return 1;
    if (!isfinite(a) || !isfinite(b)) return 0;
    double rtol = 1.4901161193847656e-08;
    return fabs(a - b) <= rtol * fmax(fabs(a), fabs(b));
}

static double accum_norm(double acc, double v, int pnorm)
{
    if (pnorm == 1) return acc + v;
    if (pnorm == 2) return acc + v * v;
    return v > acc ? v : acc;
}

/* common.jl:54-77: violation vector of rows then variables, then its p-norm */
double ora_norm_violations(int64_t m, int64_t n, const double *E, const double *gL,
                           const double *gU, const double *x, const double *xL,
                           const double *xU, int pnorm)
{
    double acc = 0.0;
    for (int64_t i = 0; i < m; ++i) {
        double v = 0.0;
        if (E[i] > gU[i]) v = E[i] - gU[i];
        else if (E[i] < gL[i]) v = gL[i] - E[i];
        acc = accum_norm(acc, v, pnorm);
    }
    for (int64_t j = 0; j < n; ++j) {
        double v = 0.0;
        if (x[j] > xU[j]) v = x[j] - xU[j];
        else if (x[j] < xL[j]) v = xL[j] - x[j];
        acc = accum_norm(acc, v, pnorm);
    }
    return pnorm == 2 ? sqrt(acc) : acc;
}

/* common.jl:14-23 */
double ora_kt_residuals(int64_t m, int64_t n, const double *df, const double *lambda,
                        const double *mult_x_U, const double *mult_x_L,
                        const int64_t *colptr, const int64_t *rowval, const double *nzval)
{
    double res = 0.0, scalar = 1.0;
    double *rown = (double *)calloc((size_t)(m > 0 ? m : 1), sizeof(double));
    for (int64_t j = 0; j < n; ++j) {
        double jtl = 0.0;
        for (int64_t k = colptr[j]; k < colptr[j + 1]; ++k) {
            jtl += nzval[k] * lambda[rowval[k]];
            rown[rowval[k]] += nzval[k] * nzval[k];
        }
        double v = fabs(df[j] + jtl + mult_x_U[j] - mult_x_L[j]);
        if (v > res) res = v;
        scalar = fmax(scalar, fabs(df[j]));
        scalar = fmax(scalar, fabs(mult_x_U[j]));
        scalar = fmax(scalar, fabs(mult_x_L[j]));
    }
    for (int64_t i = 0; i < m; ++i)
        scalar = fmax(scalar, fabs(lambda[i]) * sqrt(rown[i]));
    free(rown);
    return res / scalar;
}

/* common.jl:30-47 */
double ora_norm_complementarity(int64_t m, const double *E, const double *gL, const double *gU,
                                const double *lambda, int pnorm)
{
    double acc = 0.0, denom = 0.0;
    for (int64_t i = 0; i < m; ++i) {
        double c = 0.0;
        if (gL[i] != gU[i]) {
            c = fmin(E[i] - gL[i], gU[i] - E[i]) * lambda[i];
            denom += lambda[i] * lambda[i];
        }
        acc = accum_norm(acc, fabs(c), pnorm);
    }
    if (pnorm == 2) acc = sqrt(acc);
    return acc / (1.0 + sqrt(denom));
}

/* merit.jl:15 (scalar mu, vector violation) */
double ora_compute_derivative(double dfp, double mu, int64_t m, const double *cons_viol)
{
    double s = 0.0;
    for (int64_t i = 0; i < m; ++i) s += cons_viol[i];
    return dfp - mu * s;
}

/* sqp.jl:190-213 (compute_derivative(sqp)) on top of merit.jl:13-17: the whole directional derivative from the
 * SQP state.  mu_vec != NULL selects the vector-penalty forms (merit.jl:14,17: mu' * cons_viol), else the scalar ones.
 * feasibility_restoration: dfp is the sum of the slack values of the last sub-problem (slack[0..nslack)), and
 * cons_viol[i] is the violation of E[i] - viol[i] -- zero up to rounding, restated literally. */
double ora_compute_derivative_full(int64_t n, int64_t m, const double *df, const double *p, const double *E,
                                   const double *gL, const double *gU, double mu, const double *mu_vec,
                                   int feasibility_restoration, const double *slack, int64_t nslack)
{
    double dfp = 0.0, acc = 0.0;
    if (feasibility_restoration) {
        for (int64_t k = 0; k < nslack; ++k) dfp += slack[k];
        for (int64_t i = 0; i < m; ++i) {
            double viol = fmax(0.0, fmax(E[i] - gU[i], gL[i] - E[i]));
            double lhs = E[i] - viol;
            double cv = fmax(0.0, fmax(lhs - gU[i], gL[i] - lhs));
            acc += mu_vec ? mu_vec[i] * cv : cv;
        }
    } else {
        for (int64_t j = 0; j < n; ++j) dfp += df[j] * p[j];
        for (int64_t i = 0; i < m; ++i) {
            double cv = fmax(0.0, fmax(E[i] - gU[i], gL[i] - E[i]));
            acc += mu_vec ? mu_vec[i] * cv : cv;
        }
    }
    return mu_vec ? dfp - acc : dfp - mu * acc;
}

/* sqp_line_search.jl:270-294: compute_mu_rule1! / rule2! / rule3! (vector penalty mu[m], updated in place).
 * viol1 = norm_violations(sqp, 1), dfp = df'p, half_pHp = 0.5 p'Hp. */
void ora_compute_mu_rule(int rule, int64_t iter, double rho, double viol1, double dfp, double half_pHp, int64_t m,
                         const double *lambda, double *mu)
{
    double denom = fmax((1.0 - rho) * viol1, 1.0e-8);
    double t = (dfp + fmax(half_pHp, 0.0)) / denom;
    for (int64_t i = 0; i < m; ++i) {
        if (rule == 1) { mu[i] = fmax(mu[i], t); mu[i] = fmax(mu[i], fabs(lambda[i])); }
        else if (rule == 2) { if (iter == 1) mu[i] = t; else mu[i] = fmax(mu[i], fabs(lambda[i])); }
        else mu[i] = fmax(mu[i], fabs(lambda[i]));
    }
}

/* sqp_line_search.jl:303-334 */
double ora_armijo_alpha(double phi0, double dir_deriv, double pnorm_inf, double tol_direction,
                        double eta, double tau, double min_alpha,
                        double (*phi_at)(void *, double), void *ud, int *valid)
{
    double alpha = 1.0;
    *valid = 1;
    if (pnorm_inf <= tol_direction) return alpha;
    double phi = phi_at(ud, alpha);
    while (phi > phi0 + eta * alpha * dir_deriv) {
        if (alpha < min_alpha) { *valid = 0; break; }
        alpha *= tau;
        phi = phi_at(ud, alpha);
    }
    return alpha;
}
