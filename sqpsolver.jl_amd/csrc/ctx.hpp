// ctx.hpp -- the library context: device memory layout for a batch of NLP instances that share
// dimensions and sparsity, and the per-instance state machines of the interior-point method and
// of the SQP-TR outer loop.  Everything numerical lives in HBM; the host only sequences kernels.
#pragma once
#include <memory>
#include <vector>
#include "../../include/sqphip.h"
#include "../../include/sqphip_test_hooks.h"
#include "sqphip_internal.hpp"
#include "sparse.hpp"

namespace sqphip {

// interior-point phases of one instance (kernels act only on instances in the phase they serve)
// PH_SOLVE: first solve behind a factorisation (its forward half is fused into the factorisation); PH_RESOLVE: a
// further right-hand side through the same factors -- the corrector of the predictor-corrector mode or a refinement step
// Transitions on a side stream (Ctx::side_on, ipm_sweep): the kernels that end a sub-problem, run the stage of run! and start the
// next sub-problem then work CONCURRENTLY with the factorisation / solve / post kernels of the same group, on instances the
// main kernels do not touch.  The two sides hand instances over at the end of a sweep only (k_sqp_count, behind an event of
// each stream): the main side files a finished interior-point run as PH_DONE2 (-> PH_DONE there), the side files a started
// sub-problem with its first right-hand side as PH_PEND (-> PH_FACTOR there) and uses PH_PREP_S where the main side uses
// PH_PREP, so that neither side ever sees a state of the other half-written.
enum { PH_IDLE = 0, PH_PREP = 1, PH_FACTOR = 2, PH_SOLVE = 3, PH_STEP = 4, PH_MPC = 5, PH_RESOLVE = 6, PH_DONE = 9,
       PH_PEND = 10, PH_DONE2 = 11, PH_PREP_S = 12 };
enum { ROW_FREE = 0, ROW_EQ = 1, ROW_INEQ = 2 };

struct IpmState {
    // request
    int mode, start, stage;        // stage 0 = programme itself, 1 = phase-1 feasibility check
    double delta, mu_pen;
    // set-up
    double sf, rho_big, soft_w, hsc;
    // iteration
    double mu, tau, dw, dw_last, dw_floor, cn, relres, rn, e0;
    int iter, rc, fac_attempt, dir_attempt, refine_it, n_acc, n_acc2, n_acc3;
    int mpc, use_soc;              // predictor-corrector mode of this solve; second-order terms valid for the step
    double cavg;                   // average complementarity at the top of the iteration
    // outcome
    int prev_mode;                 // 1 + mode of this instance's last solved sub-problem (options.ipm_warm_start), 0 none
    int status, ipm_iters, n_factor, n_solve;   // n_solve: forward + backward solves with the factors (incl. refinement)
    int reload;                    // flat products (DV::flat): the working vector of the solves is to be loaded from 1 wN (refinement), 2 rhs (corrector)
    int sel;                       // which of the two factorisations of the last sweep the solves use (mfront.hip, candidates)
    int acc_rule;                  // how the last run of this sub-problem ended: 0 scaled error <= ipm_tol, 1 / 2 / 3 the acceptable-termination rules (b_ipm_prepare)
    double elastic;
};

#define SQPHIP_QLOG_CAP 64
// SQP-TR state of one instance: the scalar fields of SqpTR
// (/root/reference/src/algorithms/sqp_trust_region.jl:6-24, sqp.jl:16-59)
struct SqpState {
    double f, phi, mu, Delta, prim_infeas, dual_infeas, pnorm, obj_val;
    double f_trial, phi_k, q0, ared, pred;
    int iter, ret, step_acceptance, fr, sub_status, done, stage, need_qp, qp_mode, want_eval;
    int n_qp, trace_len, it_ipm, soc_pending, lp_pending, started;
    long tot_ipm, tot_fac, tot_sol;
    long md_qp[4], md_ipm[4], md_fac[4];   // the same work split by sub-problem mode (0 QP, 1 FR, 2 SOC, 3 LP phase)
    int qlog_n, qlog[4 * SQPHIP_QLOG_CAP]; // the last sub-problems of this instance: mode, MOI status, iterations, factorisations
    float qerr[SQPHIP_QLOG_CAP];           // ... their final scaled error (IpmState.e0) and the rule that ended them (IpmState.acc_rule)
    signed char qrule[SQPHIP_QLOG_CAP];
    long term_rule[4];                     // sub-problems ended by the tolerance / acceptable rule 1, 2, 3 since sqphip_sqp_reset
    int budget;          // outer iterations this instance may still start in the current sqp_run call
};

#define SQPHIP_TRACE_COLS 12
#define SQPHIP_TRACE_CAP 4096

// device view of the multifrontal plan (sparse.hpp, MfPlan) and of the front arena
struct MfDev {
    int ns;
    const int *first, *nc, *nr, *rowptr, *rows, *rel, *child_ptr, *child;
    const long *off;
    long stride;                          // doubles of front storage per instance
    double *fronts;                       // [B][stride]
    double *fronts1, *vals1;              // second candidate of a sweep (speculative next shift, k_inertia); null: off
    const int *asm_ptr, *dest_loc, *dest_rc, *item_ptr;
    const MfItem *items;
    const int *ea_ptr, *ea_rc, *ea_src_ptr, *ea_src;
    const int *ev_ptr, *ev_idx, *ev_src_ptr, *ev_src;
    const MfFrontDesc *desc;              // packed per-front records
    const MfGather *ea_ent, *ev_ent;      // packed gather entries (extend-add of fronts / of update vectors)
    const int *level_ptr, *level_sn;      // supernodes by level of the assembly tree (leaves first)
    int nlevels, max_front;
    const int *sched, *sol_items;
    // the top of the assembly tree for k_mf_solve_top2 (sparse.hpp MfTopFront; top_n == 0: not in use)
    const MfTopFront *top_fr;
    const int *top_gptr, *top_gsrc, *top_rows, *top_ext;
    int top_n, top_next, top_utotal, top_xtotal, top_buf0, top_buf1;
    // the spine of the factorisation for k_mf_spine (sparse.hpp MfSpineFront; sp_n == 0: level launches throughout)
    const MfSpineFront *sp_fr;
    const MfGather *sp_ent;
    const int *sp_src, *sp_rel;
    int sp_n, sp_T, sp_stage;             // fronts, tiles of the tallest, doubles of the staging area
    int nnzK;                             // destinations (structural entries of the lower triangle)
    double *vals;                         // [B][nnzK] assembled values of the destinations (k_mf_values)
};

// Scenario queue (sqphip_sqp_stream_*): more scenarios than slots.  A slot whose run has terminated stores its result
// under its scenario id, takes the next id from a device-wide counter, loads that scenario's data from the tables and
// starts over -- inside the stage kernel, no host involvement; the batch stays full until the queue is empty.
struct StreamDev {
    int M;                                            // scenarios (0: no queue)
    int *next;                                        // position in qids of the next scenario to hand out
    int *qend;                                        // [1] positions < *qend are valid (the queue may grow or shrink between runs)
    int *qids;                                        // [M] scenario ids in hand-out order (identity after _begin)
    int *slot_scen;                                   // [B] scenario a slot works on; -2 fresh slot, -1 queue exhausted
    const double *xL, *xU, *gL, *gU, *ohm, *c2, *c1, *x0;   // [M][.] scenario tables
    double *rx, *robj;                                // results: final point [M][n], objective
    int *rstat, *riter;                               // ... run! status (src/status.jl), iterations
};

// everything kernels need, by value
struct DV {
    int n, m, nlin, N, Npad, ld, B;       // N = n + m, Npad = stride of the full-length vectors rhs / sol / wN
    int Nf, Fpad;                         // order of the factorised matrix (N, or n + mk condensed) and its padding:
                                          // K is Fpad x Fpad (ld = Fpad); xv / vv / dinv have stride Fpad
    int condense, mk;                     // options.kkt_condense; number of kept (gL == gU) rows
    const int *kpos, *krow;               // row -> position among the kept rows or -1; kept position -> row
    // order of the factorised matrix (options.kkt_tile_order; identity otherwise): unknown u = variable j or
    // n + (kept-row position | row) -> position; position -> unknown or -1 (identity padding); Ts = leading tile
    // columns that are mutually independent
    const int *upos, *uinv;
    int Ts;
    const unsigned char *tmask;           // [remainder tile][leading tile] coupling mask (null: none), order.hip
    int sparse;                           // 1: multifrontal LDL^T of the sparse matrix (mfront.hip); K is not allocated
    MfDev mf;
    int nnzj_coo, nnzh_coo, nnzjc, nnzhc;
    // shared structure
    const int *jcolptr, *jrowval, *jrowptr, *jrcol, *jrslot;
    const int *hcolptr, *hrowval;
    const int *jg_ptr, *jg_src, *hg_ptr, *hg_src;
    // per-instance NLP bounds
    double *xL, *xU, *gL, *gU;
    // per-instance QP request
    double *xk, *cin, *bE, *jcoo, *hcoo, *jv, *hv;
    // canonical programme
    double *c, *hd, *lb, *ub, *lo, *hi, *wp, *wm;
    int *rtype, *rbase, *hard;
    // iterate, directions, work vectors
    double *p, *zl, *zu, *s, *tp, *tm, *y, *vl, *vu, *zp, *zm, *rdir;
    double *dp, *dzl, *dzu, *ds, *dtp, *dtm, *dy, *dvl, *dvu;
    double *rd, *rp, *sigp, *Dd, *rhs, *sol, *wn, *wN;
    double *socZL, *socZU, *socZP, *socZM, *socVL, *socVU;   // predictor's dz*dx per complementarity pair
    // linear algebra
    double *K, *dinv, *xv, *vv;
    double *dinv1, *vv1;                  // pivots / D^-1 L^-1 b of the second candidate (sparse path)
    int spec_mode;                        // which first shifts get a second candidate: 1 shrink attempts and retries, 2 retries only
    // QP outputs
    double *op, *olam, *omxU, *omxL, *oslack;
    IpmState *ist;
    int *phase;
    int *counters;      // [0] instances iterating, [1] start flags, [2] SQP not done, [3] start flags
    double ipm_tol;
    int ipm_max_iter, ipm_phase1, ipm_corrector, ipm_warm;
    int side;                             // 0: transitions in line; 1: this launch is a transition kernel on the side stream; 2: a main kernel beside it
    double refine_tol;                    // refinement step when the relative residual is above this (condensed form)
    int vstage;                           // doubles of dynamic LDS of the vector stages (n + N; 0: the vectors do not fit, ipm.hip)
    int hfull;                            // 1: the Hessian pattern is completely dense (n^2 entries of the full symmetric CSC): hess_row reads column k at row j (ipm.hip)
    int vals_inline;                      // 1: the stage kernel behind the Newton right-hand side assembles the matrix values too (mf_values_block)
    int flat;                             // 1: the sparse products of the vector stages by flat kernels over the batch (large instances, ipm.hip)
    double *fH, *fJt, *fJ, *fX;           // ... their results: H v and J' w [B][n], J v and J x of the eliminated rows [B][m]
    // ---- SQP level
    double *x, *lambda, *mxL, *mxU, *df, *E, *pstep, *psoc, *plam, *pmxL, *pmxU, *Esoc, *tmpx, *tmpE,
        *hlam;
    SqpState *sst;
    double *trace;      // [B][CAP][COLS]
    // ---- ACOPF evaluator data
    int nb, ng, nl, ref_bus;
    StreamDev stream;
    int dense_nlp;                        // 1: the synthetic dense-Hessian NLP (acopf_dev.hpp dense_eval; sqphip_dense_attach)
    const double *dnQ, *dnA;              // ... its shared Q [n][n] and A [m][n]
    double *dnc; double dn_kappa;         // ... per-instance linear cost [B][n]; weight of the quartic term
    int acr;                              // 1: rectangular voltage coordinates (acopf_dev.hpp acr_eval), 0: polar
    int acwr, nbp;                        // 1: W-space form of examples/acopf/acwr.jl (acwr_eval); its bus pairs i < j
    const int *bp_i, *bp_j, *br_bp;       // pair -> buses; branch -> pair
    const double *br_sig, *bp_tmin, *bp_tmax;   // branch orientation in its pair (+-1); tan of the pairs' angle limits
    int ndc; const double *dc_loss1;   // HVDC lines (shared): 4 variables each behind all others, one loss row each at the end
    int nsh; const int *sh_bus, *sh_of_bus; const double *sh_gs, *sh_bs;   // bus shunts (shared): list, bus -> index or -1
    const int *f_bus, *t_bus, *gen_bus, *bal_ptr, *bal_colP, *bal_colQ;
    const double *bal_coef;
    double *br_ohm, *c2, *c1, *x0;   // per instance; br_ohm[inst][nl][12] = (A, Bc, Bs) of p_f, q_f, p_t, q_t
    // options
    double tol_direction, tol_residual, tol_infeas, init_mu, tr_size;
    int max_iter, use_soc, literal_quirks;
};

// phase codes by the side a kernel runs on (see the enum)
static __device__ __forceinline__ int ph_done(const DV &d) { return d.side == 2 ? PH_DONE2 : PH_DONE; }
static __device__ __forceinline__ int ph_prep(const DV &d) { return d.side == 1 ? PH_PREP_S : PH_PREP; }
static __device__ __forceinline__ int ph_fact(const DV &d) { return d.side == 1 ? PH_PEND : PH_FACTOR; }

struct Ctx {
    sqphip_options opt;
    DV d;                       // device view (pointers into the arenas below)
    LdltPlan plan;
    std::shared_ptr<MfPlan> mfp_;   // multifrontal plan (d.sparse), shared with the lanes
    const MfPlan &mfp() const { return *mfp_; }
    // Instance groups ("lanes") of the batched SQP run: contiguous sub-batches, each with its own HIP stream, pinned
    // counter slots and timers, driven by its own host thread inside sqphip_sqp_run.  A lane is a shallow Ctx whose
    // device view points into the owner's arenas at the group's first instance (it owns no device memory).  The
    // level-by-level launches of one group leave most of the chip idle near the top of the assembly tree; kernels of
    // another group's stream fill it (measured +12 % QP/s with four groups on 512 x IEEE-118).
    std::vector<std::unique_ptr<Ctx>> lanes;
    bool is_lane = false, owns_stream = true;
    long mf_factor_launches = 0, n_sweeps = 0;
    // Transitions between sub-problems run every trans_period-th sweep of a RUN (ipm_sweep): run_sweep counts the sweeps
    // of the current sqphip_sqp_run / _stream_run call, so the first sweep of every run is a transition sweep -- slots
    // armed by the scenario queue draw their scenario there, whatever the lifetime counter n_sweeps says.
    long run_sweep = 0;
    bool post_split = false;        // experiment (SQPHIP_POST_SPLIT): the right-hand side of the next iteration by its own launch behind k_ipm_post
    bool want_resolve = true;       // monotone rule: this sweep carries the second solve slot (an instance asked for a refinement solve)
    // transitions on a side stream (see the PH_ enum): side_mode = allowed on this context (SQPHIP_SIDE_TRANS), side_on = in
    // use by the current run; evS[k & 3]: side job of sweep k done, evC[k & 3]: hand-over kernel of sweep k done
    bool side_mode = false, side_on = false;
    hipStream_t side = nullptr;
    hipEvent_t evS[4] = {}, evC[4] = {};
    // second shift per sweep in the TAIL of a run only (large batches, where it costs throughput while every instance is busy):
    // on while at most spec_tail instances of the group have work left (sqp_run_lane); 0: off.  spec_mode0: the mode at creation
    int spec_tail = 0, spec_mode0 = 0;
    int trans_period = 0;           // 0: by group size (3 from 64 instances, 2 from 32, else 1); SQPHIP_TRANS_PERIOD, read at creation
    Timers tm;
    std::vector<void *> allocs;
    std::string err;
    hipStream_t stream = nullptr;
    int *h_counters = nullptr;  // pinned
    std::vector<int> h_kpos;    // host copy of DV::kpos (row -> kept position or -1)
    bool acopf_attached = false;
    bool mf_big_lds = false;        // the multifrontal kernels were granted 160 KB of dynamic LDS on this context's device (mf_device_setup)
    bool stream_started = false;    // scenario queue: the slots have been armed (sqphip_sqp_stream_run / _run_some)
    // RCCL communicator for the status gather (comm.hip); null: single rank
    void *comm = nullptr, *comm_buf = nullptr;
    int comm_world = 1, comm_rank = 0, comm_cap = 0;   // comm_cap: block capacity comm_buf was sized for
    // host copies of structure for misc use
    int64_t n = 0, m = 0;
    // counters
    int64_t n_qp = 0, n_ipm_iter = 0, n_factor = 0, n_solve = 0;
    double total_seconds = 0;
    int last_ipm_iters = 0, last_n_factor = 0, last_rule = -1;
    double last_e0 = 0.0;

    template <class T> T *dalloc(size_t count)
    {
        void *p = nullptr;
        SQPHIP_HIP_OK(hipMalloc(&p, sizeof(T) * (count ? count : 1)));
        SQPHIP_HIP_OK(hipMemsetAsync(p, 0, sizeof(T) * (count ? count : 1), stream));
        allocs.push_back(p);
        return (T *)p;
    }
    template <class T> T *upload(const std::vector<T> &v)
    {
        T *p = dalloc<T>(v.size());
        if (!v.empty())
            SQPHIP_HIP_OK(hipMemcpyAsync(p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, stream));
        return p;
    }
    ~Ctx();
};

// ipm.hip
void ipm_run_all(Ctx &C);            // runs every instance whose IpmState.start is set, to completion
void ipm_sweep(Ctx &C, bool sqp_level);
void sqp_stage_kernels(Ctx &C, hipStream_t s, const DV &d);      // sqp.hip: SQP-level kernels of a sweep (on the stream and with the device view given)
void sqp_stream_arm(Ctx &C);          // scenario queue: every slot draws its first scenario in the first sweep
void sqp_stream_rearm(Ctx &C);        // ... slots that found the queue empty look again (ids were appended)
void launch_qp_gather(Ctx &C);       // COO -> CSC for instances with start set (stage 0)
// comm.hip
void comm_release(Ctx &C);
// mfront.hip
void mf_device_setup(Ctx &C);
void mf_factor(Ctx &C, int want, bool with_rhs, bool values_done = false);
void mf_solve(Ctx &C, int want, bool skip_fwd, bool inertia = false);   // inertia: the streamed top kernel tests the inertia of PH_FACTOR instances first
bool mf_solve_tests_inertia(const Ctx &C);     // ... which it can when the plan has a streamed top (k_mf_solve_top2)
// acopf.hip
void launch_acopf_eval_point(Ctx &C, int inst, const double *x_dev, double sigma, const double *lam_dev,
                             double *f_dev, double *grad_dev, double *g_dev, double *jcoo_dev,
                             double *hcoo_dev);
// sqp.hip
void sqp_reset(Ctx &C);
void sqp_run(Ctx &C, int max_outer);
void merit_eval(Ctx &C, int op, double a0, double a1, int flag, double *out_host);
void armijo_eval(Ctx &C, int inst, double mu, double phi0, double D, double eta, double tau, double min_alpha, int fr,
                 double *out3_host);

}  // namespace sqphip

struct sqphip_ctx { sqphip::Ctx c; };     // the opaque handle of include/sqphip.h
