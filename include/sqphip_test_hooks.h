/* sqphip_test_hooks.h -- entry points of libsqphip.so that exist for the parity tests and the micro-benchmarks only.
 * NOT part of the drop-in boundary (include/sqphip.h): no caller of the AbstractSubOptimizer seat needs them, and the
 * Julia shim binds none of them.  Kept in the shared library so that tests reach the kernels through the same C ABI. */
#ifndef SQPHIP_TEST_HOOKS_H
#define SQPHIP_TEST_HOOKS_H
#include "sqphip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Host-only test hook (no GPU, never on the product path): builds the multifrontal plan of the structure and runs a
 * plain host reference of its numeric phase -- assembly from the item lists, front-by-front partial LDL^T with the
 * right-hand side carried along, backward substitution -- on the Newton matrix
 *     [ hsc H + diag(hd + sigp + dw + 1e-8) + J_I' (D_I + 1e-8)^-1 J_I    J_K' ;  J_K   -(D_K + 1e-8) ]
 * (rows with rtype 0 are free: diagonal -1, no coupling).  Jval / Hval in the COO order of the structure; Dd, rtype
 * per row; sigp, hd per variable; rhs / sol / dinv_by_unknown in unknown order (variables, then kept rows);
 * npos = positive pivots.  CPU tests compare it with a dense solve to validate the plan the kernels run. */
int sqphip_mf_host_solve(int64_t n, int64_t m, int64_t nnzJ, const int64_t *jrow, const int64_t *jcol,
                         int64_t nnzH, const int64_t *hrow, const int64_t *hcol, const double *gL, const double *gU,
                         int32_t condense, const double *Jval, const double *Hval, const double *Dd,
                         const double *sigp, const double *hd, const int32_t *rtype, double hsc, double dw,
                         const double *rhs, double *sol, double *dinv_by_unknown, int32_t *npos);
/* After sqphip_mf_host_solve: relative error of the host replay of the streamed top-of-tree solve (k_mf_solve_top2) from
 * its own plan arrays against the plain recursion of the same call; -1 when the plan has no such top (a front of more than
 * 128 rows or 84 columns, or SQPHIP_MF_TOP2=0).  Not thread safe (one global). */
double sqphip_mf_host_top2_err(void);
/* ... and of the host replay of the spine kernel's front assembly (k_mf_spine: gather entries from the arena, the block the
 * previous front hands over through its row map, destination list) against the images the plain recursion assembled; -1
 * when the plan has no spine (a front of more than eight tiles near the top, or SQPHIP_MF_SPINE=0). */
double sqphip_mf_host_spine_err(void);
/* Device twin of sqphip_mf_host_solve (kernel-level parity tests): the same Newton matrix, assembled, factorised and
 * solved by the multifrontal kernels in instance `inst` of a context that uses the sparse solver (kkt_mode 2, or 0
 * where it selects it).  sol_fused: right-hand side carried through the factorisation; sol_standalone: the
 * stand-alone forward / backward kernels on the same factors.  Leaves the instance idle. */
int sqphip_mf_solve_test(sqphip_ctx *ctx, int32_t inst, const double *Jval, const double *Hval, const double *Dd,
                         const double *sigp, const double *hd, const int32_t *rtype, double hsc, double dw,
                         const double *rhs, double *sol_fused, double *sol_standalone, double *dinv_by_unknown);
/* ---- kernel-level entry points (parity tests, micro-benchmarks) -------------------------------
 * Batched dense LDL^T without pivoting of `batch` symmetric N x N matrices given as full
 * column-major host arrays A[batch][N*N] (lower triangle read).  On return L (unit lower) is in the
 * strict lower triangle, dinv[batch][N] = 1/D.  npos[batch] = number of positive pivots. */
int sqphip_ldlt_factor_host(int32_t device, int32_t batch, int64_t N, double *A, double *dinv,
                            int32_t *npos);
/* Factor + solve K x = rhs for each batch member; x overwrites rhs[batch][N]. */
int sqphip_ldlt_solve_host(int32_t device, int32_t batch, int64_t N, const double *A,
                           double *rhs);
/* time `reps` factorisations of resident random quasi-definite matrices; returns seconds per
 * factorisation of the whole batch and seconds spent in the trailing-update kernel */
int sqphip_ldlt_bench(int32_t device, int32_t batch, int64_t N, int32_t reps,
                      double *sec_per_factor, double *sec_trailing, int64_t *trailing_launches);

/* on-box fp64 MFMA issue-rate probe (register-resident v_mfma_f64_16x16x4_f64 loop), TFLOP/s */
/* test hook: factorise random batches with and without the look-ahead schedule and count repetitions whose
 * factors differ in any bit (must be 0) */
int sqphip_ldlt_stress(int32_t device, int32_t batch, int64_t N, int32_t reps, int32_t *mismatches);
int sqphip_mfma_f64_peak(int32_t device, double *tflops);

#ifdef __cplusplus
}
#endif
#endif
