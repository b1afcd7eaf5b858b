/* c_abi_smoke.c -- a plain C caller of libsqphip.so: dlopen, no Python, no C++.
 *
 * Proves what a foreign host (the Julia ccall shim of julia/SqpHip.jl) relies on: the struct layouts of
 * include/sqphip.h as a C compiler sees them, the symbol names, and the call sequence of the sub-problem seat.
 *   c_abi_smoke <path to libsqphip.so>        host-only part (no GPU): option defaults by field, counters struct size,
 *                                             sqphip_tr_update, sqphip_kkt_symbolic on the toy structure
 *   c_abi_smoke <path to libsqphip.so> gpu    ... plus the toy NLP of /root/reference/test/ext_solver.jl:14-28 at its
 *                                             start point through sqphip_create / sqphip_qp_solve (mode QP is
 *                                             infeasible there, mode FR solves: SURVEY.md section 8c KAT-3)
 * Exit code 0 = all checks passed.  Built and run by tests/test_abi.py and tests/test_gpu_parity.py with gcc. */
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/sqphip.h"
#include "../include/sqphip_test_hooks.h"

#define SYM(name) __typeof__(&name) p_##name = (__typeof__(&name))dlsym(h, #name); \
    if (!p_##name) { fprintf(stderr, "missing symbol %s\n", #name); return 2; }
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "check failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: %s libsqphip.so [gpu]\n", argv[0]); return 2; }
    void *h = dlopen(argv[1], RTLD_NOW);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
    SYM(sqphip_default_options) SYM(sqphip_tr_update) SYM(sqphip_kkt_symbolic) SYM(sqphip_create) SYM(sqphip_qp_solve)
    SYM(sqphip_qp_stats) SYM(sqphip_destroy) SYM(sqphip_last_error) SYM(sqphip_get_counters) SYM(sqphip_gather_status)

    /* struct layout: 10 doubles, 3 int32 (+ 4 bytes padding), 1 double, 8 int32 */
    CHECK(sizeof(sqphip_options) == 10 * 8 + 3 * 4 + 4 + 8 + 8 * 4);
    CHECK(sizeof(sqphip_counters) == 29 * 8);                      /* 29 eight-byte fields */
    sqphip_options o;
    memset(&o, 0xff, sizeof o);
    p_sqphip_default_options(&o);
    CHECK(o.tol_direction == 1e-8 && o.tol_residual == 1e-8 && o.tol_infeas == 1e-8);    /* parameters.jl:17-19 */
    CHECK(o.init_mu == 1.0 && o.max_mu == 1e10 && o.tr_size == 10.0 && o.rho == 0.8 && o.eta == 0.4 && o.tau == 0.9);
    CHECK(o.min_alpha == 1e-6 && o.max_iter == 3000 && o.use_soc == 0 && o.literal_quirks == 1);
    CHECK(o.ipm_tol == 1e-9 && o.ipm_max_iter == 200 && o.ipm_phase1 == 0 && o.device == 0 && o.ipm_corrector == 0);
    CHECK(o.kkt_condense == 1 && o.kkt_tile_order == 1 && o.kkt_mode == 0 && o.ipm_warm_start == 0);

    /* ratio test / radius update, sqp_trust_region.jl:529-538, :574-577 */
    int32_t acc; double dn;
    CHECK(p_sqphip_tr_update(1.0, 2.0, 10.0, 10.0, 1e8, 1e-8, &acc, &dn) == 0 && acc == 1 && dn == 20.0);
    CHECK(p_sqphip_tr_update(-1.0, 2.0, 10.0, 4.0, 1e8, 1e-8, &acc, &dn) == 0 && acc == 0 && dn == 2.0);

    /* toy NLP of test/ext_solver.jl: n = 2, m = 4; rows: X >= -2 (linear), X^2 - X - 2 = 0, X Y - 1 = 0, X Y >= 0 */
    int64_t jrow[] = {1, 2, 3, 3, 4, 4}, jcol[] = {1, 1, 1, 2, 1, 2};
    int64_t hrow[] = {1, 2}, hcol[] = {1, 1};
    double gL[] = {-2, 0, 0, 0}, gU[] = {INFINITY, 0, 0, INFINITY};
    double xL[] = {-INFINITY, -INFINITY}, xU[] = {INFINITY, INFINITY};
    sqphip_symbolic_stats st;
    int32_t pos[6];
    CHECK(p_sqphip_kkt_symbolic(2, 4, 6, jrow, jcol, 2, hrow, hcol, gL, gU, 1, 1, 0, -1.0, pos, &st) == 0);
    CHECK(st.order == 4 && st.n_supernodes >= 1 && st.nnz_l >= 1);      /* 2 variables + 2 equality rows */
    CHECK(pos[2] > pos[0] && pos[3] > pos[0] && pos[3] > pos[1]);       /* rows behind the variables they couple to */

    if (argc > 2 && strcmp(argv[2], "gpu") == 0) {
        sqphip_ctx *ctx = NULL;
        CHECK(p_sqphip_create(&ctx, 2, 4, 1, 6, jrow, jcol, 2, hrow, hcol, xL, xU, gL, gU, &o, 1) == 0 && ctx);
        /* x0 = (0, 0): f = X^2 + X -> df = (1, 0); E = (0, -2, -1, 0); J values in COO order; H(lambda = 0) = [[2, 0], [0, 0]] */
        double x0[] = {0, 0}, df[] = {1, 0}, E[] = {0, -2, -1, 0}, Jv[] = {1, -1, 0, 0, 0, 0}, Hv[] = {2, 0};
        double p[2], lam[4], mU[2], mL[2], slack[8];
        int32_t status = -1, its = 0, nf = 0;
        CHECK(p_sqphip_qp_solve(ctx, SQPHIP_MODE_QP, x0, 10.0, 1.0, df, E, Jv, Hv, p, lam, mU, mL, slack, &status) == 0);
        CHECK(status == SQPHIP_MOI_LOCALLY_INFEASIBLE && p[0] == 0.0 && lam[1] == 0.0);   /* row 3 reads 0 = 1 */
        CHECK(p_sqphip_qp_solve(ctx, SQPHIP_MODE_FR, x0, 10.0, 1.0, df, E, Jv, Hv, p, lam, mU, mL, slack, &status) == 0);
        CHECK(status == SQPHIP_MOI_LOCALLY_SOLVED && fabs(p[0] + 2.0) < 1e-6);             /* KAT-3: p1 = -2 */
        double mass = 0; for (int i = 0; i < 8; ++i) mass += slack[i];
        CHECK(fabs(mass - 1.0) < 1e-6);                                                    /* FR LP optimum 1 */
        CHECK(p_sqphip_qp_stats(ctx, &its, &nf) == 0 && its > 0 && nf >= its);
        sqphip_counters c;
        CHECK(p_sqphip_get_counters(ctx, &c) == 0 && c.n_qp == 2 && c.kkt_order >= 4);
        int32_t r1, i1, d1;
        CHECK(p_sqphip_gather_status(ctx, 1, &r1, &i1, &d1) == 0);                         /* single rank: local table */
        CHECK(p_sqphip_gather_status(ctx, 2, &r1, &i1, &d1) != 0 && strlen(p_sqphip_last_error(ctx)) > 0);
        p_sqphip_destroy(ctx);
        printf("c_abi_smoke: gpu part ok (FR step p1 = %.9f, %d interior-point iterations)\n", p[0], its);
    }
    printf("c_abi_smoke: ok\n");
    return 0;
}
