// api.hip -- the extern "C" boundary of libsqphip.so (declared in include/sqphip.h).
// Builds the shared sparsity structure once on the host (what `sparse(I,J,V)` does in
// /root/reference/src/algorithms/sqp_trust_region.jl:47-48,56-57 plus the symmetric mirroring of
// sqp.jl:96-101), lays the batch out in HBM and forwards to the kernels.
#include "ctx.hpp"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <tuple>

using namespace sqphip;

Ctx::~Ctx()
{
    lanes.clear();              // before the arenas they point into
    tm.flush();
    comm_release(*this);
    plan.destroy_lookahead();
    for (void *p : allocs) hipFree(p);
    if (h_counters) hipHostFree(h_counters);
    if (side) { hipStreamDestroy(side); for (auto &e : evS) if (e) hipEventDestroy(e); for (auto &e : evC) if (e) hipEventDestroy(e); }
    if (stream && owns_stream) hipStreamDestroy(stream);
}

namespace {

struct Pattern {
    std::vector<int> colptr, rowval, g_ptr, g_src;     // CSC + per-slot gather lists over COO indices
};

Pattern build_pattern(int64_t ncols, int64_t nnz, const int64_t *row, const int64_t *col, bool sym)
{
    std::vector<std::tuple<int64_t, int64_t, int64_t>> e;   // (col, row, coo index)
    e.reserve(sym ? 2 * nnz : nnz);
    for (int64_t k = 0; k < nnz; ++k) {
        e.emplace_back(col[k] - 1, row[k] - 1, k);
        if (sym && row[k] != col[k]) e.emplace_back(row[k] - 1, col[k] - 1, k);
    }
    std::sort(e.begin(), e.end());
    Pattern P;
    P.colptr.assign(ncols + 1, 0);
    P.g_ptr.push_back(0);
    for (size_t i = 0; i < e.size(); ++i) {
        const bool fresh = i == 0 || std::get<0>(e[i]) != std::get<0>(e[i - 1]) ||
                           std::get<1>(e[i]) != std::get<1>(e[i - 1]);
        if (fresh) {
            if (i) P.g_ptr.push_back((int)P.g_src.size());
            P.rowval.push_back((int)std::get<1>(e[i]));
            P.colptr[std::get<0>(e[i]) + 1]++;
        }
        P.g_src.push_back((int)std::get<2>(e[i]));
    }
    P.g_ptr.push_back((int)P.g_src.size());
    if (e.empty()) P.g_ptr.assign(1, 0);
    for (int64_t j = 0; j < ncols; ++j) P.colptr[j + 1] += P.colptr[j];
    return P;
}

template <class F> int guarded(sqphip_ctx *h, F &&f)
{
    try {
        SQPHIP_HIP_OK(hipSetDevice(h->c.opt.device));
        return f(h->c);
    } catch (const std::string &e) {
        h->c.err = e;
        return SQPHIP_EHIP;
    } catch (const std::bad_alloc &) {
        h->c.err = "out of host memory";
        return SQPHIP_ENOMEM;
    }
}

void h2d(Ctx &C, double *dst, const double *src, size_t k)
{
    if (src && k) SQPHIP_HIP_OK(hipMemcpyAsync(dst, src, sizeof(double) * k, hipMemcpyHostToDevice, C.stream));
}
void d2h(Ctx &C, double *dst, const double *src, size_t k)
{
    if (dst && k) SQPHIP_HIP_OK(hipMemcpyAsync(dst, src, sizeof(double) * k, hipMemcpyDeviceToHost, C.stream));
}

// the device view of instances [i0, i0 + Bg): every per-instance array advanced to the group's first instance, so that
// the kernels see an ordinary (smaller) batch.  Shared structure arrays stay as they are.
DV group_view(const DV &d, int i0, int Bg, int g)
{
    DV v = d;
    v.B = Bg;
    const long n = d.n, m = d.m, o = i0;
    double **nv[] = { &v.xL, &v.xU, &v.xk, &v.cin, &v.c, &v.hd, &v.lb, &v.ub, &v.p, &v.zl, &v.zu, &v.dp, &v.dzl, &v.dzu, &v.rd,
                      &v.sigp, &v.wn, &v.op, &v.omxU, &v.omxL, &v.x, &v.mxL, &v.mxU, &v.df, &v.pstep, &v.psoc, &v.pmxL,
                      &v.pmxU, &v.tmpx, &v.x0, &v.socZL, &v.socZU };
    for (auto pp : nv) *pp += o * n;
    double **mv[] = { &v.gL, &v.gU, &v.bE, &v.lo, &v.hi, &v.wp, &v.wm, &v.s, &v.tp, &v.tm, &v.y, &v.vl, &v.vu, &v.ds, &v.dtp,
                      &v.dtm, &v.dy, &v.dvl, &v.dvu, &v.rp, &v.Dd, &v.olam, &v.lambda, &v.E, &v.plam, &v.Esoc, &v.tmpE,
                      &v.hlam, &v.zp, &v.zm, &v.rdir, &v.socZP, &v.socZM, &v.socVL, &v.socVU };
    for (auto pp : mv) *pp += o * m;
    v.oslack += 2 * o * m;
    v.rtype += o * m; v.rbase += o * m; v.hard += o * m;
    v.jcoo += o * d.nnzj_coo; v.hcoo += o * d.nnzh_coo; v.jv += o * d.nnzjc; v.hv += o * d.nnzhc;
    v.rhs += o * d.Npad; v.sol += o * d.Npad; v.wN += o * d.Npad;
    if (v.flat) { v.fH += o * n; v.fJt += o * n; v.fJ += o * m; v.fX += o * m; }
    v.xv += o * d.Fpad; v.vv += o * d.Fpad; v.dinv += o * d.Fpad;
    v.ist += o; v.sst += o; v.phase += o;
    if (v.stream.slot_scen) v.stream.slot_scen += o;
    v.trace += o * SQPHIP_TRACE_CAP * SQPHIP_TRACE_COLS;
    v.counters = d.counters + 8 * (g + 1);
    if (v.br_ohm) { v.br_ohm += o * d.nl * 12; v.c2 += o * d.ng; v.c1 += o * d.ng; }
    if (v.dnc) v.dnc += o * n;
    v.mf.fronts += o * d.mf.stride; v.mf.vals += o * (long)d.mf.nnzK;
    if (v.mf.fronts1) { v.mf.fronts1 += o * d.mf.stride; v.mf.vals1 += o * (long)d.mf.nnzK; v.dinv1 += o * d.Fpad; v.vv1 += o * d.Fpad; }
    return v;
}

// (re)build the lanes from the owner's current device view (after sqphip_create and after sqphip_acopf_attach)
void make_lanes(Ctx &C)
{
    C.lanes.clear();
    int G = 1;
    // (round 3, with the shorter solve kernels: four groups from 64 instances on -- 64 resident scenarios 1 955 -> 2 015 QP/s;
    //  SQPHIP_GROUP_RULE=<g32>,<g16>: groups for batches of 32 .. 63 and 16 .. 31, experiment)
    if (C.d.sparse) {
        int g32 = 4, g16 = 2;           // (32 resident scenarios: 1 034 / 1 074 / 1 119 QP/s with one / two / four groups)
        if (const char *e = getenv("SQPHIP_GROUP_RULE")) sscanf(e, "%d,%d", &g32, &g16);
        G = C.d.B >= 64 ? 4 : (C.d.B >= 32 ? g32 : (C.d.B >= 16 ? g16 : 1));
    }
    if (const char *e = getenv("SQPHIP_GROUPS")) G = atoi(e);
    if (G > C.d.B) G = C.d.B;
    if (G > 7) G = 7;                       // counter slots
    if (!C.d.sparse || G < 2) return;
    for (int g = 0; g < G; ++g) {
        const int lo = (int)((long)C.d.B * g / G), hi = (int)((long)C.d.B * (g + 1) / G);
        std::unique_ptr<Ctx> L(new Ctx());
        L->is_lane = true;
        L->opt = C.opt; L->n = C.n; L->m = C.m; L->acopf_attached = C.acopf_attached;
        L->mfp_ = C.mfp_;
        L->trans_period = C.trans_period; L->mf_big_lds = C.mf_big_lds; L->post_split = C.post_split; L->side_mode = C.side_mode; L->spec_tail = C.spec_tail; L->spec_mode0 = C.spec_mode0;
        L->d = group_view(C.d, lo, hi - lo, g);
        // the first group runs on the owner's stream (idle during sqphip_sqp_run): HIP maps streams onto four hardware
        // queues by default, and a fifth stream would share one -- measured: 3131 QP/s with five streams against 5216
        // with four (or with GPU_MAX_HW_QUEUES=8)
        // (experiment, SQPHIP_CU_PARTITION = 1 | 2: every group on a stream restricted to a quarter of the CUs -- 1: a contiguous
        //  quarter of the mask bits, 2: every G-th bit -- so that one group's throughput-bound launches cannot slow another group's
        //  latency-bound ones; measured: profiles/r04_ab_experiments.txt)
        static const int cu_part = getenv("SQPHIP_CU_PARTITION") ? atoi(getenv("SQPHIP_CU_PARTITION")) : 0;
        if (cu_part != 0) {
            hipDeviceProp_t prop; SQPHIP_HIP_OK(hipGetDeviceProperties(&prop, C.opt.device));
            const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
            std::vector<uint32_t> mask(words, 0u);
            for (int c = 0; c < ncu; ++c) {
                const bool mine = cu_part == 1 ? (c * G / ncu == g) : (c % G == g);
                if (mine) mask[c / 32] |= 1u << (c % 32);
            }
            SQPHIP_HIP_OK(hipExtStreamCreateWithCUMask(&L->stream, (uint32_t)words, mask.data()));
        } else
        if (g == 0) { L->stream = C.stream; L->owns_stream = false; }
        else SQPHIP_HIP_OK(hipStreamCreateWithFlags(&L->stream, getenv("SQPHIP_STREAM_BLOCKING") ? hipStreamDefault : hipStreamNonBlocking));
        SQPHIP_HIP_OK(hipHostMalloc((void **)&L->h_counters, 8 * sizeof(int)));
        L->tm.enabled = C.tm.enabled;
        C.lanes.push_back(std::move(L));
    }
}

__global__ void k_qp_request(DV d, int inst, int mode, double delta, double mu_pen)
{
    for (int i = threadIdx.x; i < d.B; i += blockDim.x) d.ist[i].start = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        IpmState &I = d.ist[inst];
        I.mode = mode; I.delta = delta; I.mu_pen = mu_pen;
        I.stage = 0; I.rho_big = 1e4; I.start = 1; I.ipm_iters = 0; I.n_factor = 0; I.n_solve = 0; I.status = 0;
    }
}

// test hook (sqphip_mf_solve_test): instance `inst` alone enters phase `ph` with the given Hessian scale / delta_w
__global__ void k_mf_test_setup(DV d, int inst, double hsc, double dw, int ph)
{
    for (int i = threadIdx.x; i < d.B; i += blockDim.x) { d.phase[i] = PH_IDLE; d.ist[i].start = 0; }
    __syncthreads();
    if (threadIdx.x == 0) { d.ist[inst].hsc = hsc; d.ist[inst].dw = dw; d.ist[inst].stage = 0; d.ist[inst].sel = 0; d.ist[inst].fac_attempt = 0; d.phase[inst] = ph; }
}

}  // namespace

extern "C" void sqphip_default_options(sqphip_options *o)
{
    // /root/reference/src/parameters.jl:17-29
    o->tol_direction = 1e-8; o->tol_residual = 1e-8; o->tol_infeas = 1e-8;
    o->init_mu = 1.0; o->max_mu = 1e10; o->tr_size = 10.0;
    o->rho = 0.8; o->eta = 0.4; o->tau = 0.9; o->min_alpha = 1e-6;
    o->max_iter = 3000; o->use_soc = 0; o->literal_quirks = 1;
    o->ipm_tol = 1e-9; o->ipm_max_iter = 200; o->ipm_phase1 = 0; o->device = 0; o->ipm_corrector = 0; o->kkt_condense = 1; o->kkt_tile_order = 1;
    o->kkt_mode = 0; o->ipm_warm_start = 0;
}

extern "C" int sqphip_create(sqphip_ctx **out, int64_t n, int64_t m, int64_t num_linear, int64_t nnzJ,
                             const int64_t *jrow, const int64_t *jcol, int64_t nnzH, const int64_t *hrow,
                             const int64_t *hcol, const double *xL, const double *xU, const double *gL,
                             const double *gU, const sqphip_options *opt, int32_t batch)
{
    if (!out || n <= 0 || m < 0 || batch <= 0 || !opt || num_linear < 0 || num_linear > m) return SQPHIP_EINVAL;
    if (nnzJ < 0 || nnzH < 0 || (nnzJ > 0 && (!jrow || !jcol)) || (nnzH > 0 && (!hrow || !hcol))) return SQPHIP_EINVAL;
    if (!xL || !xU || (m > 0 && (!gL || !gU))) return SQPHIP_EINVAL;
    for (int64_t k = 0; k < nnzJ; ++k)
        if (jrow[k] < 1 || jrow[k] > m || jcol[k] < 1 || jcol[k] > n) return SQPHIP_EINVAL;
    for (int64_t k = 0; k < nnzH; ++k)
        if (hrow[k] < 1 || hrow[k] > n || hcol[k] < 1 || hcol[k] > n) return SQPHIP_EINVAL;
    // rows unbounded on both sides create no constraint upstream and shift its indexing
    // (SURVEY.md App. C #15): rejected
    for (int64_t i = 0; i < m; ++i)
        if (gL[i] == -INFINITY && gU[i] == INFINITY) return SQPHIP_EINVAL;
    sqphip_ctx *h = new (std::nothrow) sqphip_ctx();
    if (!h) return SQPHIP_ENOMEM;
    h->c.opt = *opt;
    int rc = guarded(h, [&](Ctx &C) {
        SQPHIP_HIP_OK(hipStreamCreate(&C.stream));
        SQPHIP_HIP_OK(hipHostMalloc((void **)&C.h_counters, 8 * sizeof(int)));
        DV &d = C.d;
        std::memset(&d, 0, sizeof(d));
        C.n = n; C.m = m;
        const int B = batch;
        d.n = (int)n; d.m = (int)m; d.nlin = (int)num_linear; d.N = (int)(n + m);
        d.Npad = (d.N + 63) / 64 * 64; d.B = B;
        d.nnzj_coo = (int)nnzJ; d.nnzh_coo = (int)nnzH;
        Pattern PJ = build_pattern(n, nnzJ, jrow, jcol, false);
        Pattern PH = build_pattern(n, nnzH, hrow, hcol, true);
        d.nnzjc = (int)PJ.rowval.size(); d.nnzhc = (int)PH.rowval.size();
        d.hfull = n > 1 && (long)d.nnzhc == (long)n * n;
        // CSR view of J
        std::vector<int> rptr(m + 1, 0), rcol(d.nnzjc), rslot(d.nnzjc);
        for (int s = 0; s < d.nnzjc; ++s) rptr[PJ.rowval[s] + 1]++;
        for (int64_t i = 0; i < m; ++i) rptr[i + 1] += rptr[i];
        {
            std::vector<int> fill(rptr.begin(), rptr.end() - 1);
            for (int j = 0; j < (int)n; ++j)
                for (int s = PJ.colptr[j]; s < PJ.colptr[j + 1]; ++s) {
                    const int i = PJ.rowval[s];
                    rcol[fill[i]] = j; rslot[fill[i]] = s; fill[i]++;
                }
        }
        d.jcolptr = C.upload(PJ.colptr); d.jrowval = C.upload(PJ.rowval);
        d.jrowptr = C.upload(rptr); d.jrcol = C.upload(rcol); d.jrslot = C.upload(rslot);
        {   // condensed form: the equality rows (and rows too long to eliminate, sparse.hpp) stay in the factorised
            // matrix, all others are eliminated
            std::vector<int> kpos(m > 0 ? m : 1, -1), krow;
            for (int64_t i = 0; i < m; ++i)
                if (kkt_row_is_kept(gL[i], gU[i], rptr[i + 1] - rptr[i])) { kpos[i] = (int)krow.size(); krow.push_back((int)i); }
            d.condense = opt->kkt_condense != 0; d.mk = (int)krow.size();
            if (krow.empty()) krow.push_back(0);
            d.kpos = C.upload(kpos); d.krow = C.upload(krow);
            C.h_kpos = kpos;
        }
        {   // linear solver and order of the factorised matrix
            const std::vector<int> &kpos = C.h_kpos;
            const int nu = d.condense ? d.n + d.mk : d.N;       // unknowns of the factorised system
            std::vector<int> upos(nu);
            d.Ts = 0; d.Nf = nu; d.sparse = 0;
            if (opt->kkt_mode != 1) {
                // multifrontal plan of the sparse matrix (full form: every row is its own unknown)
                std::vector<int> kp(kpos);
                if (!d.condense) for (int64_t i = 0; i < m; ++i) kp[i] = (int)i;
                SymOptions so;
                if (const char *e = getenv("SQPHIP_MF_SMALL_FRONT")) so.small_front = atoi(e);
                if (const char *e = getenv("SQPHIP_MF_ZERO_FRAC")) so.zero_frac = atof(e);
                if (const char *e = getenv("SQPHIP_MF_ROWS_AFTER")) so.rows_after_vars = atoi(e);
                so.merge_tiles = B <= 64 ? 4 : 0;        // (small batches: one launch per level of small fronts, mfplan.hip)
                C.mfp_ = std::make_shared<MfPlan>(mf_build_plan(d.n, (int)m, kp, d.condense ? d.mk : (int)m, PH.colptr, PH.rowval, rptr, rcol, rslot, so));
                const SparseSym &S = C.mfp().S;
                // auto: the sparse factorisation when it does a quarter of the dense work or less and no front
                // outgrows what one workgroup eliminates in reasonable time; a dense Hessian goes to the MFMA path
                const double dense_flops = (double)nu * nu * nu / 3.0;
                const bool fits = S.max_front <= 1024;
                if (opt->kkt_mode == 2 && !fits) throw std::string("kkt_mode = 2: a front of " + std::to_string(S.max_front) + " rows exceeds the multifrontal kernels' limit (1024)");
                d.sparse = opt->kkt_mode == 2 || (fits && S.max_front <= 512 && S.flops <= 0.25 * dense_flops);
            }
            if (d.sparse) {
                const MfPlan &P = C.mfp();
                const SparseSym &S = P.S;
                upos = S.pos;
                MfDev &M = d.mf;
                M.ns = S.ns; M.stride = P.stride;
                M.first = C.upload(S.sn_first); M.nc = C.upload(S.sn_nc); M.nr = C.upload(S.sn_nr);
                M.rowptr = C.upload(S.sn_rowptr); M.rows = C.upload(S.sn_rows.empty() ? std::vector<int>(1, 0) : S.sn_rows);
                M.rel = C.upload(S.rel.empty() ? std::vector<int>(1, 0) : S.rel);
                M.child_ptr = C.upload(S.child_ptr); M.child = C.upload(S.child.empty() ? std::vector<int>(1, 0) : S.child);
                M.off = C.upload(P.off);
                M.asm_ptr = C.upload(P.asm_ptr); M.dest_loc = C.upload(P.dest_loc); M.item_ptr = C.upload(P.item_ptr);
                M.items = C.upload(P.items); M.sched = C.upload(P.sched);
                auto nz = [](const std::vector<int> &v) { return v.empty() ? std::vector<int>(1, 0) : v; };
                M.dest_rc = C.upload(P.dest_rc);
                M.ea_ptr = C.upload(P.ea_ptr); M.ea_rc = C.upload(nz(P.ea_rc));
                M.ea_src_ptr = C.upload(P.ea_src_ptr); M.ea_src = C.upload(nz(P.ea_src));
                M.ev_ptr = C.upload(P.ev_ptr); M.ev_idx = C.upload(nz(P.ev_idx));
                M.ev_src_ptr = C.upload(P.ev_src_ptr); M.ev_src = C.upload(nz(P.ev_src));
                M.level_ptr = C.upload(S.level_ptr); M.level_sn = C.upload(S.level_sn);
                M.nlevels = S.nlevels; M.max_front = S.max_front;
                M.sol_items = C.upload(nz(P.sol_items));
                M.desc = C.upload(P.desc);
                M.ea_ent = C.upload(P.ea_ent.empty() ? std::vector<MfGather>(1) : P.ea_ent);
                M.ev_ent = C.upload(P.ev_ent.empty() ? std::vector<MfGather>(1) : P.ev_ent);
                int lds_max = 0;                 // LDS a workgroup may have on this device (gfx950: 160 KB)
                if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, C.opt.device) != hipSuccess) lds_max = 0;
                M.top_n = P.top2_lds_bytes > 0 && P.top2_lds_bytes <= lds_max ? (int)P.top_fr.size() : 0;
                if (M.top_n > 0) {
                    M.top_fr = C.upload(P.top_fr); M.top_gptr = C.upload(P.top_gptr); M.top_gsrc = C.upload(nz(P.top_gsrc));
                    M.top_rows = C.upload(nz(P.top_rows)); M.top_ext = C.upload(nz(P.top_ext));
                    M.top_next = (int)P.top_ext.size(); M.top_utotal = P.top_utotal; M.top_xtotal = P.top_xtotal;
                    M.top_buf0 = P.top_buf0; M.top_buf1 = P.top_buf1;
                }
                M.sp_n = P.spine_lds_bytes > 0 && P.spine_lds_bytes <= lds_max ? (int)P.sp_fr.size() : 0;
                if (M.sp_n > 0) {
                    M.sp_fr = C.upload(P.sp_fr); M.sp_ent = C.upload(P.sp_ent.empty() ? std::vector<MfGather>(1) : P.sp_ent);
                    M.sp_src = C.upload(nz(P.sp_src)); M.sp_rel = C.upload(nz(P.sp_rel)); M.sp_T = P.spine_T; M.sp_stage = P.spine_stage;
                }
                M.nnzK = (int)P.nnzK;
                M.vals = C.dalloc<double>((size_t)B * P.nnzK);
                M.fronts = C.dalloc<double>((size_t)B * P.stride);
                // Second candidate of a sweep (the speculative next shift, k_inertia / dev_util.hpp): a failed first shift then
                // costs no extra sweep.  It pays while a sweep is bound by the latency of its kernel chain -- up to ~256
                // resident instances (+15 % QP/s at 64, +8 % at 128, +6.5 % at 256) -- and costs once the front kernels are
                // throughput-bound (-5 % at 512).  SQPHIP_MF_SPEC: 0 off, 1 shrink attempts and retries, 2 retries only.
                // Large matrices (1354 / 9241 buses) rarely fail a shrink attempt and gain nothing (measured): the second
                // arena is only spent where it is small (<= 1 GB).
                const bool small_arena = (double)B * (double)P.stride * 8.0 <= 1e9;
                // (round 4, monotone sweeps -- shorter, so a saved sweep is worth less against the doubled front work: 256
                //  resident scenarios 6 541 QP/s with it, 6 627 without; 128: 4 307 / 4 110; on up to 192 instances)
                d.spec_mode = getenv("SQPHIP_MF_SPEC") ? atoi(getenv("SQPHIP_MF_SPEC")) : (B <= (opt->ipm_corrector ? 256 : 192) && small_arena ? 1 : 0);
                // ... and for larger batches in the TAIL of a run: once most instances of a group have used up their outer iterations
                // the sweeps are latency-bound again and the stragglers' failed shifts are the critical path (sqp_run_lane switches
                // it on while at most spec_tail instances of the group have work left; SQPHIP_MF_SPEC_TAIL, 0 = never).  Measured,
                // driver's command: threshold 0 / 16 / 32 / 48 / 64 / 96 -> 8 992 / 9 036 / 9 052 / 9 031 / 9 049 / 9 016 QP/s at 512 resident
                // scenarios (1.4 % fewer sweeps, identical work counters), 6 614 / 6 682 / 6 708 / 6 692 / 6 552 / 6 561 at 256
                C.spec_tail = d.spec_mode == 0 && small_arena ? (getenv("SQPHIP_MF_SPEC_TAIL") ? atoi(getenv("SQPHIP_MF_SPEC_TAIL")) : (getenv("SQPHIP_MF_SPEC") ? 0 : 32)) : 0;
                C.spec_mode0 = d.spec_mode;
                if (d.spec_mode != 0 || C.spec_tail > 0) {
                    M.vals1 = C.dalloc<double>((size_t)B * P.nnzK);
                    M.fronts1 = C.dalloc<double>((size_t)B * P.stride);
                }
            } else if (d.condense && opt->kkt_tile_order) {
                KktOrder o = kkt_order(d.n, (int)m, kpos, d.mk, PH.colptr, PH.rowval, rptr, rcol, /*rows_last=*/true);
                upos = o.pos; d.Ts = o.Ts; d.Nf = o.Nf;
                if (o.Ts > 0 && o.Tr > 0 && !getenv("SQPHIP_NO_TILE_MASK")) {
                    C.plan.tmask = C.upload(o.tmask);
                    d.tmask = C.plan.tmask;
                    C.plan.pair_ptr = C.upload(o.pair_ptr);
                    C.plan.pair_k = C.upload(o.pair_k.empty() ? std::vector<int>(1, 0) : o.pair_k);
                    // per tile pair: (entries of the lower triangle in the tile) x 2 x 64 x (sub-panels in its list)
                    double fl = 0.0;
                    for (int ti = 0, pi = 0; ti < o.Tr; ++ti)
                        for (int tj = 0; tj <= ti; ++tj, ++pi)
                            fl += (ti == tj ? 64.0 * 65.0 / 2.0 : 64.0 * 64.0) * 2.0 * 64.0 * (o.pair_ptr[pi + 1] - o.pair_ptr[pi]);
                    C.plan.lead_update_flops = fl;
                }
            } else {
                for (int u = 0; u < nu; ++u) upos[u] = u;
            }
            d.Fpad = (d.Nf + 63) / 64 * 64; d.ld = d.Fpad;
            std::vector<int> uinv(d.Fpad, -1);
            for (int u = 0; u < nu; ++u) uinv[upos[u]] = u;
            d.upos = C.upload(upos); d.uinv = C.upload(uinv);
        }
        d.hcolptr = C.upload(PH.colptr); d.hrowval = C.upload(PH.rowval);
        d.jg_ptr = C.upload(PJ.g_ptr); d.jg_src = C.upload(PJ.g_src);
        d.hg_ptr = C.upload(PH.g_ptr); d.hg_src = C.upload(PH.g_src);
        const size_t Bn = (size_t)B * n, Bm = (size_t)B * m, BN = (size_t)B * d.Npad;
        d.xL = C.dalloc<double>(Bn); d.xU = C.dalloc<double>(Bn);
        d.gL = C.dalloc<double>(Bm); d.gU = C.dalloc<double>(Bm);
        for (int b = 0; b < B; ++b) {
            h2d(C, d.xL + (size_t)b * n, xL, n); h2d(C, d.xU + (size_t)b * n, xU, n);
            h2d(C, d.gL + (size_t)b * m, gL, m); h2d(C, d.gU + (size_t)b * m, gU, m);
        }
        d.xk = C.dalloc<double>(Bn); d.cin = C.dalloc<double>(Bn); d.bE = C.dalloc<double>(Bm);
        d.jcoo = C.dalloc<double>((size_t)B * nnzJ); d.hcoo = C.dalloc<double>((size_t)B * nnzH);
        d.jv = C.dalloc<double>((size_t)B * d.nnzjc); d.hv = C.dalloc<double>((size_t)B * d.nnzhc);
        double **nvec[] = { &d.c, &d.hd, &d.lb, &d.ub, &d.p, &d.zl, &d.zu, &d.dp, &d.dzl, &d.dzu, &d.rd,
                            &d.sigp, &d.wn, &d.op, &d.omxU, &d.omxL, &d.x, &d.mxL, &d.mxU, &d.df, &d.pstep,
                            &d.psoc, &d.pmxL, &d.pmxU, &d.tmpx, &d.x0, &d.socZL, &d.socZU };
        for (auto pp : nvec) *pp = C.dalloc<double>(Bn);
        double **mvec[] = { &d.lo, &d.hi, &d.wp, &d.wm, &d.s, &d.tp, &d.tm, &d.y, &d.vl, &d.vu, &d.ds,
                            &d.dtp, &d.dtm, &d.dy, &d.dvl, &d.dvu, &d.rp, &d.Dd, &d.olam, &d.lambda, &d.E,
                            &d.plam, &d.Esoc, &d.tmpE, &d.hlam, &d.zp, &d.zm, &d.rdir,
                            &d.socZP, &d.socZM, &d.socVL, &d.socVU };
        for (auto pp : mvec) *pp = C.dalloc<double>(Bm);
        d.oslack = C.dalloc<double>(2 * Bm);
        d.rtype = C.dalloc<int>(Bm); d.rbase = C.dalloc<int>(Bm); d.hard = C.dalloc<int>(Bm);
        d.rhs = C.dalloc<double>(BN); d.sol = C.dalloc<double>(BN); d.wN = C.dalloc<double>(BN);
        const size_t BF = (size_t)B * d.Fpad;          // solve vectors and pivots live in the factorised order
        d.xv = C.dalloc<double>(BF); d.vv = C.dalloc<double>(BF); d.dinv = C.dalloc<double>(BF);
        if (d.sparse && d.mf.fronts1) { d.vv1 = C.dalloc<double>(BF); d.dinv1 = C.dalloc<double>(BF); }
        d.K = C.dalloc<double>(d.sparse ? 1 : (size_t)B * d.ld * d.Fpad);
        d.ist = C.dalloc<IpmState>(B); d.sst = C.dalloc<SqpState>(B);
        d.phase = C.dalloc<int>(B); d.counters = C.dalloc<int>(64);
        d.trace = C.dalloc<double>((size_t)B * SQPHIP_TRACE_CAP * SQPHIP_TRACE_COLS);
        d.ipm_tol = opt->ipm_tol; d.ipm_max_iter = opt->ipm_max_iter; d.ipm_phase1 = opt->ipm_phase1; d.ipm_corrector = opt->ipm_corrector; d.ipm_warm = opt->ipm_warm_start;
        d.refine_tol = getenv("SQPHIP_REFINE_TOL") ? atof(getenv("SQPHIP_REFINE_TOL")) : 1e-11;
        d.vstage = (d.n + d.N <= 7000 && !getenv("SQPHIP_NO_VSTAGE")) ? d.n + d.N : 0;      // up to 56 KB of LDS per workgroup
        // large instances (the vectors do not fit the LDS of one workgroup): the sparse products of the vector stages by flat
        // kernels over the whole batch (ipm.hip, k_sp_products); SQPHIP_VEC_FLAT=1 / 0 forces / forbids it (tests, A / B runs)
        d.flat = getenv("SQPHIP_VEC_FLAT") ? (atoi(getenv("SQPHIP_VEC_FLAT")) != 0) : (d.vstage == 0 && d.n + d.N > 7000);
        // (measured, round 4: the stage kernel assembling the values of its instance -- eight destinations per thread, each a
        //  chain item list -> items -> operands -- takes longer than the flat k_mf_values launch it saves: 7 528 against 7 706
        //  QP/s at 512 x IEEE-118, 1 880 against 2 002 at 64, +2 % on IEEE-14; on request only)
        d.vals_inline = d.sparse && !d.flat && getenv("SQPHIP_MF_VALS_INLINE") && atoi(getenv("SQPHIP_MF_VALS_INLINE")) == 1;
        if (d.flat) {
            d.vstage = 0;
            d.fH = C.dalloc<double>(Bn); d.fJt = C.dalloc<double>(Bn); d.fJ = C.dalloc<double>(Bm); d.fX = C.dalloc<double>(Bm);
        }
        d.tol_direction = opt->tol_direction; d.tol_residual = opt->tol_residual;
        d.tol_infeas = opt->tol_infeas; d.init_mu = opt->init_mu; d.tr_size = opt->tr_size;
        d.max_iter = opt->max_iter; d.use_soc = opt->use_soc; d.literal_quirks = opt->literal_quirks;
        C.plan.N = d.Nf; C.plan.Npad = d.Fpad; C.plan.T = d.Fpad / 64; C.plan.ld = d.ld; C.plan.B = B;
        C.plan.Ts = d.Ts;
        C.plan.stream = C.stream;
        if (opt->kkt_tile_order && !d.condense && !d.sparse)
            C.err = "note: options.kkt_tile_order needs options.kkt_condense = 1 and was ignored";
        if (!d.sparse) {
            C.plan.Wbuf = C.dalloc<double>((size_t)std::max(2 * LdltPlan::MAX_R, d.Ts) * B * d.Fpad * 64);
            C.plan.init_lookahead();
        }
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        if (const char *e = getenv("SQPHIP_TRANS_PERIOD")) C.trans_period = atoi(e);      // experiment switch, read once per context
        if (d.sparse) mf_device_setup(C);
        C.post_split = getenv("SQPHIP_POST_SPLIT") && atoi(getenv("SQPHIP_POST_SPLIT")) == 1;
        C.side_mode = getenv("SQPHIP_SIDE_TRANS") && atoi(getenv("SQPHIP_SIDE_TRANS")) == 1;      // transitions on a side stream (ctx.hpp)
        make_lanes(C);
        return SQPHIP_OK;
    });
    if (rc != SQPHIP_OK) {
        fprintf(stderr, "sqphip_create: %s\n", h->c.err.c_str());
        delete h;
        return rc;
    }
    *out = h;
    return SQPHIP_OK;
}

extern "C" void sqphip_destroy(sqphip_ctx *h)
{
    if (!h) return;
    hipSetDevice(h->c.opt.device);
    delete h;
}

extern "C" const char *sqphip_last_error(const sqphip_ctx *h) { return h ? h->c.err.c_str() : "null context"; }

extern "C" int sqphip_set_bounds(sqphip_ctx *h, int32_t inst, const double *xL, const double *xU,
                                 const double *gL, const double *gU)
{
    if (!h || inst < 0 || inst >= h->c.d.B || !xL || !xU || (h->c.d.m > 0 && (!gL || !gU))) return SQPHIP_EINVAL;
    for (int i = 0; i < h->c.d.m; ++i)          // as sqphip_create: a row unbounded on both sides is no constraint upstream
        if (gL[i] == -INFINITY && gU[i] == INFINITY) { h->c.err = "sqphip_set_bounds: row " + std::to_string(i) + " is unbounded on both sides"; return SQPHIP_EINVAL; }
    return guarded(h, [&](Ctx &C) {
        DV &d = C.d;
        // condensed form: the kept-row set was fixed at creation; an instance may not turn an eliminated row into
        // an equality (its D would sit at the regularisation and 1/D in the condensed matrix at 1e8)
        if (d.condense)
            for (int i = 0; i < d.m; ++i)
                if (gL[i] == gU[i] && C.h_kpos[i] < 0) {
                    C.err = "sqphip_set_bounds: row " + std::to_string(i) + " is an equality for this instance but was "
                            "not one when the context was created (options.kkt_condense = 1 fixes the kept rows)";
                    return SQPHIP_EINVAL;
                }
        h2d(C, d.xL + (size_t)inst * d.n, xL, d.n); h2d(C, d.xU + (size_t)inst * d.n, xU, d.n);
        h2d(C, d.gL + (size_t)inst * d.m, gL, d.m); h2d(C, d.gU + (size_t)inst * d.m, gU, d.m);
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_qp_solve(sqphip_ctx *h, int32_t mode, const double *x_k, double delta, double mu,
                               const double *df, const double *E, const double *Jval, const double *Hval,
                               double *p, double *lambda, double *mult_x_U, double *mult_x_L, double *slack,
                               int32_t *moi_status)
{
    if (!h || !x_k || !Jval || !p || !lambda || !mult_x_U || !mult_x_L || !moi_status) return SQPHIP_EINVAL;
    if (mode < 0 || mode > SQPHIP_MODE_INFEAS) return SQPHIP_EINVAL;
    if (mode != SQPHIP_MODE_LP && (!df || !E)) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) {
        DV &d = C.d;
        auto t0 = std::chrono::steady_clock::now();
        h2d(C, d.xk, x_k, d.n);
        h2d(C, d.cin, df, d.n);
        h2d(C, d.bE, E, d.m);
        h2d(C, d.jcoo, Jval, d.nnzj_coo);
        if (Hval) h2d(C, d.hcoo, Hval, d.nnzh_coo);
        else SQPHIP_HIP_OK(hipMemsetAsync(d.hcoo, 0, sizeof(double) * (size_t)d.nnzh_coo, C.stream));
        hipLaunchKernelGGL(k_qp_request, dim3(1), dim3(64), 0, C.stream, d, 0, (int)mode, delta, mu);
        launch_qp_gather(C);
        ipm_run_all(C);
        IpmState st;
        SQPHIP_HIP_OK(hipMemcpyAsync(&st, d.ist, sizeof(IpmState), hipMemcpyDeviceToHost, C.stream));
        d2h(C, p, d.op, d.n); d2h(C, lambda, d.olam, d.m);
        d2h(C, mult_x_U, d.omxU, d.n); d2h(C, mult_x_L, d.omxL, d.n);
        d2h(C, slack, d.oslack, 2 * (size_t)d.m);
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        *moi_status = st.status;
        C.last_ipm_iters = st.ipm_iters; C.last_n_factor = st.n_factor;
        C.last_rule = st.rc == 0 ? st.acc_rule : -1; C.last_e0 = st.e0;
        C.n_qp += 1; C.n_ipm_iter += st.ipm_iters; C.n_factor += st.n_factor; C.n_solve += st.n_solve;
        C.total_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return SQPHIP_OK;
    });
}

// Kernel-level test hook of the multifrontal path (mfront.hip) -- the device twin of sqphip_mf_host_solve: Newton
// matrix from the given values in instance `inst` of a context created with the sparse solver, factorised with the
// right-hand side fused in (sol_fused), then solved again through the stand-alone forward / backward kernels
// (sol_standalone).  Vectors in unknown order (variables, then kept rows).
extern "C" int sqphip_mf_solve_test(sqphip_ctx *h, int32_t inst, const double *Jval, const double *Hval,
                                    const double *Dd, const double *sigp, const double *hd, const int32_t *rtype,
                                    double hsc, double dw, const double *rhs, double *sol_fused,
                                    double *sol_standalone, double *dinv_by_unknown)
{
    if (!h || inst < 0 || inst >= h->c.d.B || !Jval || !Dd || !sigp || !hd || !rtype || !rhs) return SQPHIP_EINVAL;
    if (!h->c.d.sparse) { h->c.err = "sqphip_mf_solve_test: the context does not use the sparse solver"; return SQPHIP_ESTATE; }
    return guarded(h, [&](Ctx &C) {
        DV &d = C.d;
        const SparseSym &S = C.mfp().S;
        const size_t on = (size_t)inst * d.n, om = (size_t)inst * d.m;
        h2d(C, d.jcoo + (size_t)inst * d.nnzj_coo, Jval, d.nnzj_coo);
        if (Hval) h2d(C, d.hcoo + (size_t)inst * d.nnzh_coo, Hval, d.nnzh_coo);
        else SQPHIP_HIP_OK(hipMemsetAsync(d.hcoo + (size_t)inst * d.nnzh_coo, 0, sizeof(double) * (size_t)d.nnzh_coo, C.stream));
        hipLaunchKernelGGL(k_qp_request, dim3(1), dim3(64), 0, C.stream, d, (int)inst, 0, 1.0, 1.0);
        launch_qp_gather(C);                       // COO -> CSC values of the instance (start flag set, stage 0)
        hipLaunchKernelGGL(k_mf_test_setup, dim3(1), dim3(64), 0, C.stream, d, (int)inst, hsc, dw, (int)PH_FACTOR);
        h2d(C, d.Dd + om, Dd, d.m); h2d(C, d.sigp + on, sigp, d.n); h2d(C, d.hd + on, hd, d.n);
        SQPHIP_HIP_OK(hipMemcpyAsync(d.rtype + om, rtype, sizeof(int) * (size_t)d.m, hipMemcpyHostToDevice, C.stream));
        std::vector<double> xp(d.Fpad, 0.0), out(d.Fpad), dv(d.Fpad);
        for (int u = 0; u < S.nu; ++u) xp[S.pos[u]] = rhs[u];
        double *xv = d.xv + (size_t)inst * d.Fpad;
        h2d(C, xv, xp.data(), d.Fpad);
        mf_factor(C, PH_FACTOR, true);
        mf_solve(C, PH_FACTOR, true);
        d2h(C, out.data(), xv, d.Fpad); d2h(C, dv.data(), d.dinv + (size_t)inst * d.Fpad, d.Fpad);
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        for (int u = 0; u < S.nu; ++u) {
            if (sol_fused) sol_fused[u] = out[S.pos[u]];
            if (dinv_by_unknown) dinv_by_unknown[u] = dv[S.pos[u]];
        }
        h2d(C, xv, xp.data(), d.Fpad);
        mf_solve(C, PH_FACTOR, false);
        d2h(C, out.data(), xv, d.Fpad);
        hipLaunchKernelGGL(k_mf_test_setup, dim3(1), dim3(64), 0, C.stream, d, (int)inst, 0.0, 0.0, (int)PH_IDLE);
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        SQPHIP_HIP_OK(hipGetLastError());
        if (sol_standalone) for (int u = 0; u < S.nu; ++u) sol_standalone[u] = out[S.pos[u]];
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_qp_stats(const sqphip_ctx *h, int32_t *ipm_iters, int32_t *n_factor)
{
    if (!h) return SQPHIP_EINVAL;
    if (ipm_iters) *ipm_iters = h->c.last_ipm_iters;
    if (n_factor) *n_factor = h->c.last_n_factor;
    return SQPHIP_OK;
}

extern "C" int sqphip_qp_termination(const sqphip_ctx *h, int32_t *rule, double *scaled_error)
{
    if (!h) return SQPHIP_EINVAL;
    if (rule) *rule = h->c.last_rule;
    if (scaled_error) *scaled_error = h->c.last_e0;
    return SQPHIP_OK;
}

// ---- merit path -------------------------------------------------------------------------------
extern "C" int sqphip_norm_violations(sqphip_ctx *h, const double *E, const double *x, int32_t pnorm, double *out)
{
    if (!h || !E || !x || !out || pnorm < 0 || pnorm > 2) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) {
        h2d(C, C.d.E, E, C.d.m); h2d(C, C.d.x, x, C.d.n);
        merit_eval(C, 0, 0.0, 0.0, pnorm, out);
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_kt_residuals(sqphip_ctx *h, const double *df, const double *lambda, const double *mult_x_U,
                                   const double *mult_x_L, const double *Jval, double *out)
{
    if (!h || !df || !lambda || !mult_x_U || !mult_x_L || !Jval || !out) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) {
        DV &d = C.d;
        h2d(C, d.df, df, d.n); h2d(C, d.lambda, lambda, d.m); h2d(C, d.mxU, mult_x_U, d.n);
        h2d(C, d.mxL, mult_x_L, d.n); h2d(C, d.jcoo, Jval, d.nnzj_coo);
        merit_eval(C, 1, 0.0, 0.0, 0, out);
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_norm_complementarity(sqphip_ctx *h, const double *E, const double *lambda, int32_t pnorm,
                                           double *out)
{
    if (!h || !E || !lambda || !out || pnorm < 0 || pnorm > 2) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) {
        h2d(C, C.d.E, E, C.d.m); h2d(C, C.d.lambda, lambda, C.d.m);
        merit_eval(C, 2, 0.0, 0.0, pnorm, out);
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_compute_phi(sqphip_ctx *h, double f_trial, const double *E_trial, const double *x_trial,
                                  double mu, int32_t feasibility_restoration, double *phi)
{
    if (!h || !E_trial || !x_trial || !phi) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) {
        h2d(C, C.d.E, E_trial, C.d.m); h2d(C, C.d.x, x_trial, C.d.n);
        merit_eval(C, 3, f_trial, mu, feasibility_restoration, phi);
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_compute_qmodel(sqphip_ctx *h, const double *x, const double *p, const double *df,
                                     const double *E, const double *Jval, const double *Hval, double mu,
                                     int32_t with_step, double *q)
{
    if (!h || !x || !E || !q) return SQPHIP_EINVAL;
    if (with_step && (!p || !df || !Jval)) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) {
        DV &d = C.d;
        h2d(C, d.x, x, d.n); h2d(C, d.E, E, d.m);
        if (with_step) {
            h2d(C, d.pstep, p, d.n); h2d(C, d.df, df, d.n); h2d(C, d.jcoo, Jval, d.nnzj_coo);
            if (Hval) h2d(C, d.hcoo, Hval, d.nnzh_coo);
            else SQPHIP_HIP_OK(hipMemsetAsync(d.hcoo, 0, sizeof(double) * (size_t)d.nnzh_coo, C.stream));
        }
        merit_eval(C, 4, 0.0, mu, with_step, q);
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_compute_derivative(sqphip_ctx *h, const double *df, const double *p, const double *E,
                                         double mu, double *D)
{
    if (!h || !df || !p || !E || !D) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) {
        h2d(C, C.d.df, df, C.d.n); h2d(C, C.d.pstep, p, C.d.n); h2d(C, C.d.E, E, C.d.m);
        merit_eval(C, 5, 0.0, mu, 0, D);
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_compute_derivative_full(sqphip_ctx *h, const double *df, const double *p, const double *E, double mu,
                                              const double *mu_vec, int32_t feasibility_restoration, const double *slack,
                                              double *D)
{
    if (!h || !E || !D) return SQPHIP_EINVAL;
    if (feasibility_restoration ? !slack : (!df || !p)) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) {
        DV &d = C.d;
        h2d(C, d.E, E, d.m);
        if (feasibility_restoration) h2d(C, d.oslack, slack, 2 * (size_t)d.m);
        else { h2d(C, d.df, df, d.n); h2d(C, d.pstep, p, d.n); }
        if (mu_vec) h2d(C, d.plam, mu_vec, d.m);
        merit_eval(C, 6, 0.0, mu, (feasibility_restoration ? 1 : 0) | (mu_vec ? 2 : 0), D);
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_compute_mu_rule_dev(sqphip_ctx *h, int32_t rule, int64_t iter, double rho, const double *x,
                                          const double *E, const double *df, const double *p, const double *Hval,
                                          const double *lambda, double *mu)
{
    if (!h || rule < 1 || rule > 3 || !x || !E || !df || !p || !lambda || !mu) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) {
        DV &d = C.d;
        h2d(C, d.x, x, d.n); h2d(C, d.E, E, d.m); h2d(C, d.df, df, d.n); h2d(C, d.pstep, p, d.n);
        h2d(C, d.lambda, lambda, d.m); h2d(C, d.plam, mu, d.m);
        SQPHIP_HIP_OK(hipMemsetAsync(d.jcoo, 0, sizeof(double) * (size_t)d.nnzj_coo, C.stream));
        if (Hval) h2d(C, d.hcoo, Hval, d.nnzh_coo);
        else SQPHIP_HIP_OK(hipMemsetAsync(d.hcoo, 0, sizeof(double) * (size_t)d.nnzh_coo, C.stream));
        double t = 0.0;
        merit_eval(C, 7, rho, 0.0, rule | ((iter == 1 ? 1 : 0) << 4), &t);
        d2h(C, mu, d.plam, d.m);
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_acopf_armijo(sqphip_ctx *h, int32_t inst, const double *x, const double *p, double mu, double phi0,
                                   double D, double eta, double tau, double min_alpha, int32_t feasibility_restoration,
                                   double *alpha, int32_t *is_valid, int32_t *n_eval)
{
    if (!h || !h->c.acopf_attached || inst < 0 || inst >= h->c.d.B || !x || !p || !alpha || !is_valid) return SQPHIP_EINVAL;
    if (!(tau > 0.0 && tau < 1.0)) return SQPHIP_EINVAL;        // the loop must terminate
    return guarded(h, [&](Ctx &C) {
        DV &d = C.d;
        h2d(C, d.x + (size_t)inst * d.n, x, d.n); h2d(C, d.pstep + (size_t)inst * d.n, p, d.n);
        double out[3];
        armijo_eval(C, inst, mu, phi0, D, eta, tau, min_alpha, feasibility_restoration, out);
        *alpha = out[0]; *is_valid = (int32_t)out[1];
        if (n_eval) *n_eval = (int32_t)out[2];
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_tr_update(double ared, double pred, double delta, double pnorm_inf, double delta_max,
                                double tol_direction, int32_t *accept_out, double *delta_out)
{
    if (!accept_out || !delta_out) return SQPHIP_EINVAL;
    // sqp_trust_region.jl:529-538, :574-577; isapprox with rtol = sqrt(eps)
    const double rho = ared / pred;
    const bool acc = ared > 0 && rho > 0;
    double dn = delta;
    if (acc) {
        const bool approx = delta == pnorm_inf ||
                            (std::isfinite(delta) && std::isfinite(pnorm_inf) &&
                             std::fabs(delta - pnorm_inf) <=
                                 1.4901161193847656e-08 * std::fmax(std::fabs(delta), std::fabs(pnorm_inf)));
        if (approx) dn = std::fmin(2 * delta, delta_max);
    } else {
        dn = std::fmax(0.5 * std::fmin(delta, pnorm_inf), 0.1 * tol_direction);
    }
    *accept_out = acc ? 1 : 0;
    *delta_out = dn;
    return SQPHIP_OK;
}

// sqp_line_search.jl:303-334
extern "C" int sqphip_armijo_alpha(double phi0, double D, double eta, double tau, double min_alpha, double pnorm_inf,
                                   double tol_direction, sqphip_phi_fn phi, void *user, double *alpha,
                                   int32_t *is_valid)
{
    if (!phi || !alpha || !is_valid) return SQPHIP_EINVAL;
    double a = 1.0;
    int ok = 1;
    if (!(pnorm_inf <= tol_direction)) {
        double phi_x_p = phi(a, user);
        while (phi_x_p > phi0 + eta * a * D) {
            if (a < min_alpha) { ok = 0; break; }        // the step size can become too small
            a *= tau;
            phi_x_p = phi(a, user);
        }
    }
    *alpha = a;
    *is_valid = ok;
    return SQPHIP_OK;
}

// sqp_line_search.jl:270-294
extern "C" int sqphip_compute_mu_rule(int32_t rule, int64_t iter, double rho, double viol1, double dfp,
                                      double half_pHp, int64_t m, const double *lambda, double *mu)
{
    if (rule < 1 || rule > 3 || m < 0 || (m > 0 && (!lambda || !mu))) return SQPHIP_EINVAL;
    const double denom = std::fmax((1.0 - rho) * viol1, 1.0e-8);
    const double t = (dfp + std::fmax(half_pHp, 0.0)) / denom;
    for (int64_t i = 0; i < m; ++i) {
        if (rule == 1) { mu[i] = std::fmax(mu[i], t); mu[i] = std::fmax(mu[i], std::fabs(lambda[i])); }
        else if (rule == 2) mu[i] = iter == 1 ? t : std::fmax(mu[i], std::fabs(lambda[i]));
        else mu[i] = std::fmax(mu[i], std::fabs(lambda[i]));
    }
    return SQPHIP_OK;
}

// ---- ACOPF evaluator + batched SQP -----------------------------------------------------------------
// structure counts of the two layouts (acopf_synth.py acopf_layout / acr_layout) without shunts
static int acopf_nnzj(int acr, int nb, int ng, int nl, int ndc)
{
    const int nbal = 2 * nl + ng + 2 * ndc;
    return acr ? 1 + 2 * nbal + 4 * nb + 24 * nl + 2 * ndc : 32 * nl + 2 * ng + 1 + 6 * ndc;
}
static int acopf_nnzh(int acr, int nb, int ng, int nl) { return acr ? ng + 28 * nl + 4 * nb : ng + 44 * nl; }
// ... and of the W-space layout (acwr_layout): the shunt entries are part of the structure
static int acwr_n(int nb, int nbp, int ng, int nl) { return 3 * nb + 2 * nbp + 2 * ng + 4 * nl; }
static int acwr_m(int nb, int nbp, int nl) { return 1 + 3 * nb + 4 * nbp + 6 * nl; }
static int acwr_nnzj(int nb, int nbp, int ng, int nl, int ndc)
{
    const int nbal = 2 * nl + ng + 2 * ndc;
    return 1 + 2 * nbal + 2 * nb + 4 * nbp + 16 * nl + 3 * nb + 10 * nbp + 4 * nl + 2 * ndc;
}
static int acwr_nnzh(int nb, int nbp, int ng, int nl) { return ng + 4 * nl + 2 * nb + 4 * nbp; }

static int acopf_attach_impl(sqphip_ctx *h, int acr, int32_t nb, int32_t ng, int32_t nl, const int32_t *f_bus,
                             const int32_t *t_bus, const int32_t *gen_bus, const int32_t *bal_ptr,
                             const int32_t *bal_colP, const int32_t *bal_colQ, const double *bal_coef,
                             int32_t ref_bus)
{
    if (!h || nb <= 0 || ng <= 0 || nl <= 0) return SQPHIP_EINVAL;
    if (!f_bus || !t_bus || !gen_bus || !bal_ptr || !bal_colP || !bal_colQ || !bal_coef) return SQPHIP_EINVAL;
    Ctx &C0 = h->c;
    // HVDC lines add 4 variables, 1 row, 2 balance incidences (= 4 Jacobian entries) and 2 loss-row entries each
    const int n_ac = 2 * nb + 2 * ng + 4 * nl, m_ac = acr ? 1 + 4 * nb + 6 * nl : 1 + 2 * nb + 8 * nl;
    if (C0.d.n < n_ac || (C0.d.n - n_ac) % 4 != 0) return SQPHIP_EINVAL;
    const int ndc = (C0.d.n - n_ac) / 4;
    if (C0.d.m != m_ac + ndc) return SQPHIP_EINVAL;
    const int nbal = bal_ptr[nb];
    // (a structure with bus shunts has 2 (polar) / 4 (rectangular) more Jacobian and 1 / 2 more Hessian entries per
    // shunted bus at the end of the lists; sqphip_acopf_set_shunts checks the exact counts)
    const int extraJ = C0.d.nnzj_coo - acopf_nnzj(acr, nb, ng, nl, ndc), extraH = C0.d.nnzh_coo - acopf_nnzh(acr, nb, ng, nl);
    if (extraJ < 0 || extraJ != 2 * extraH || extraH > (acr ? 2 * nb : nb) || nbal != 2 * nl + ng + 2 * ndc) return SQPHIP_EINVAL;
    for (int l = 0; l < nl; ++l)
        if (f_bus[l] < 0 || f_bus[l] >= nb || t_bus[l] < 0 || t_bus[l] >= nb) return SQPHIP_EINVAL;
    for (int k = 0; k < nbal; ++k)
        if (bal_colP[k] < 0 || bal_colP[k] >= C0.d.n || bal_colQ[k] < 0 || bal_colQ[k] >= C0.d.n) return SQPHIP_EINVAL;
    if (ref_bus < 0 || ref_bus >= nb) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) {
        DV &d = C.d;
        d.nb = nb; d.ng = ng; d.nl = nl; d.ref_bus = ref_bus; d.acr = acr; d.acwr = 0;
        d.ndc = ndc;
        d.dc_loss1 = C.upload(std::vector<double>(ndc > 0 ? ndc : 1, 0.0));   // sqphip_acopf_set_dclines overrides
        d.f_bus = C.upload(std::vector<int>(f_bus, f_bus + nl));
        d.t_bus = C.upload(std::vector<int>(t_bus, t_bus + nl));
        d.gen_bus = C.upload(std::vector<int>(gen_bus, gen_bus + ng));
        d.bal_ptr = C.upload(std::vector<int>(bal_ptr, bal_ptr + nb + 1));
        d.bal_colP = C.upload(std::vector<int>(bal_colP, bal_colP + nbal));
        d.bal_colQ = C.upload(std::vector<int>(bal_colQ, bal_colQ + nbal));
        d.bal_coef = C.upload(std::vector<double>(bal_coef, bal_coef + nbal));
        d.br_ohm = C.dalloc<double>((size_t)d.B * nl * 12);
        d.c2 = C.dalloc<double>((size_t)d.B * ng); d.c1 = C.dalloc<double>((size_t)d.B * ng);
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        C.acopf_attached = true;
        make_lanes(C);
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_acopf_attach(sqphip_ctx *h, int32_t nb, int32_t ng, int32_t nl, const int32_t *f_bus,
                                   const int32_t *t_bus, const int32_t *gen_bus, const int32_t *bal_ptr,
                                   const int32_t *bal_colP, const int32_t *bal_colQ, const double *bal_coef,
                                   int32_t ref_bus)
{
    return acopf_attach_impl(h, 0, nb, ng, nl, f_bus, t_bus, gen_bus, bal_ptr, bal_colP, bal_colQ, bal_coef, ref_bus);
}

extern "C" int sqphip_acopf_attach_acr(sqphip_ctx *h, int32_t nb, int32_t ng, int32_t nl, const int32_t *f_bus,
                                       const int32_t *t_bus, const int32_t *gen_bus, const int32_t *bal_ptr,
                                       const int32_t *bal_colP, const int32_t *bal_colQ, const double *bal_coef,
                                       int32_t ref_bus)
{
    return acopf_attach_impl(h, 1, nb, ng, nl, f_bus, t_bus, gen_bus, bal_ptr, bal_colP, bal_colQ, bal_coef, ref_bus);
}

extern "C" int sqphip_acopf_attach_acwr(sqphip_ctx *h, int32_t nb, int32_t ng, int32_t nl, const int32_t *f_bus,
                                        const int32_t *t_bus, const int32_t *gen_bus, const int32_t *bal_ptr,
                                        const int32_t *bal_colP, const int32_t *bal_colQ, const double *bal_coef,
                                        int32_t ref_bus, int32_t nbp, const int32_t *bp_i, const int32_t *bp_j,
                                        const int32_t *br_bp, const double *br_sig, const double *bp_tmin,
                                        const double *bp_tmax)
{
    if (!h || nb <= 0 || ng <= 0 || nl <= 0 || nbp <= 0 || nbp > nl) return SQPHIP_EINVAL;
    if (!f_bus || !t_bus || !gen_bus || !bal_ptr || !bal_colP || !bal_colQ || !bal_coef) return SQPHIP_EINVAL;
    if (!bp_i || !bp_j || !br_bp || !br_sig || !bp_tmin || !bp_tmax) return SQPHIP_EINVAL;
    Ctx &C0 = h->c;
    const int n_ac = acwr_n(nb, nbp, ng, nl);
    if (C0.d.n < n_ac || (C0.d.n - n_ac) % 4 != 0) return SQPHIP_EINVAL;
    const int ndc = (C0.d.n - n_ac) / 4;
    if (C0.d.m != acwr_m(nb, nbp, nl) + ndc) return SQPHIP_EINVAL;
    const int nbal = bal_ptr[nb];
    if (C0.d.nnzj_coo != acwr_nnzj(nb, nbp, ng, nl, ndc) || C0.d.nnzh_coo != acwr_nnzh(nb, nbp, ng, nl) ||
        nbal != 2 * nl + ng + 2 * ndc) return SQPHIP_EINVAL;
    for (int l = 0; l < nl; ++l)
        if (f_bus[l] < 0 || f_bus[l] >= nb || t_bus[l] < 0 || t_bus[l] >= nb || br_bp[l] < 0 || br_bp[l] >= nbp ||
            !(br_sig[l] == 1.0 || br_sig[l] == -1.0)) return SQPHIP_EINVAL;
    for (int k = 0; k < nbp; ++k)
        if (bp_i[k] < 0 || bp_j[k] >= nb || bp_i[k] >= bp_j[k]) return SQPHIP_EINVAL;
    for (int k = 0; k < nbal; ++k)
        if (bal_colP[k] < 0 || bal_colP[k] >= C0.d.n || bal_colQ[k] < 0 || bal_colQ[k] >= C0.d.n) return SQPHIP_EINVAL;
    if (ref_bus < 0 || ref_bus >= nb) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) {
        DV &d = C.d;
        d.nb = nb; d.ng = ng; d.nl = nl; d.ref_bus = ref_bus; d.acr = 0; d.acwr = 1; d.nbp = nbp;
        d.ndc = ndc;
        d.dc_loss1 = C.upload(std::vector<double>(ndc > 0 ? ndc : 1, 0.0));
        d.f_bus = C.upload(std::vector<int>(f_bus, f_bus + nl));
        d.t_bus = C.upload(std::vector<int>(t_bus, t_bus + nl));
        d.gen_bus = C.upload(std::vector<int>(gen_bus, gen_bus + ng));
        d.bal_ptr = C.upload(std::vector<int>(bal_ptr, bal_ptr + nb + 1));
        d.bal_colP = C.upload(std::vector<int>(bal_colP, bal_colP + nbal));
        d.bal_colQ = C.upload(std::vector<int>(bal_colQ, bal_colQ + nbal));
        d.bal_coef = C.upload(std::vector<double>(bal_coef, bal_coef + nbal));
        d.bp_i = C.upload(std::vector<int>(bp_i, bp_i + nbp)); d.bp_j = C.upload(std::vector<int>(bp_j, bp_j + nbp));
        d.br_bp = C.upload(std::vector<int>(br_bp, br_bp + nl));
        d.br_sig = C.upload(std::vector<double>(br_sig, br_sig + nl));
        d.bp_tmin = C.upload(std::vector<double>(bp_tmin, bp_tmin + nbp));
        d.bp_tmax = C.upload(std::vector<double>(bp_tmax, bp_tmax + nbp));
        d.br_ohm = C.dalloc<double>((size_t)d.B * nl * 12);
        d.c2 = C.dalloc<double>((size_t)d.B * ng); d.c1 = C.dalloc<double>((size_t)d.B * ng);
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        C.acopf_attached = true;
        make_lanes(C);
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_acopf_set_shunts(sqphip_ctx *h, int32_t nsh, const int32_t *sh_bus, const double *gs,
                                       const double *bs)
{
    if (!h || !h->c.acopf_attached || nsh < 0) return SQPHIP_EINVAL;
    Ctx &C0 = h->c;
    const int nl = C0.d.nl, ng = C0.d.ng, nb = C0.d.nb;
    const int acr = C0.d.acr;
    if (!C0.d.acwr) {                    // (the W-space structure carries the shunt entries of every bus already)
        if (C0.d.nnzj_coo != acopf_nnzj(acr, nb, ng, nl, C0.d.ndc) + (acr ? 4 : 2) * nsh ||
            C0.d.nnzh_coo != acopf_nnzh(acr, nb, ng, nl) + (acr ? 2 : 1) * nsh) return SQPHIP_EINVAL;
        if (nsh > 0 && C0.d.nlin != (acr ? 1 : 2 * nl + 1)) return SQPHIP_EINVAL; // balance rows must not be declared linear
    }
    std::vector<int> of(nb, -1);
    for (int s = 0; s < nsh; ++s) {
        if (sh_bus[s] < 0 || sh_bus[s] >= nb || of[sh_bus[s]] >= 0) return SQPHIP_EINVAL;
        of[sh_bus[s]] = s;
    }
    return guarded(h, [&](Ctx &C) {
        DV &d = C.d;
        d.nsh = nsh;
        d.sh_bus = C.upload(std::vector<int>(sh_bus, sh_bus + nsh));
        d.sh_of_bus = C.upload(of);
        d.sh_gs = C.upload(std::vector<double>(gs, gs + nsh));
        d.sh_bs = C.upload(std::vector<double>(bs, bs + nsh));
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        make_lanes(C);
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_acopf_set_dclines(sqphip_ctx *h, int32_t ndc, const double *loss1)
{
    if (!h || !h->c.acopf_attached || ndc != h->c.d.ndc || (ndc > 0 && !loss1)) return SQPHIP_EINVAL;
    if (ndc == 0) return SQPHIP_OK;
    return guarded(h, [&](Ctx &C) {
        C.d.dc_loss1 = C.upload(std::vector<double>(loss1, loss1 + ndc));
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        make_lanes(C);
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_acopf_set_instance(sqphip_ctx *h, int32_t inst, const double *ohm, const double *c2,
                                         const double *c1, const double *x0)
{
    if (!h || !h->c.acopf_attached || inst < 0 || inst >= h->c.d.B) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) {
        DV &d = C.d;
        h2d(C, d.br_ohm + (size_t)inst * d.nl * 12, ohm, (size_t)d.nl * 12);
        h2d(C, d.c2 + (size_t)inst * d.ng, c2, d.ng); h2d(C, d.c1 + (size_t)inst * d.ng, c1, d.ng);
        h2d(C, d.x0 + (size_t)inst * d.n, x0, d.n);
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        return SQPHIP_OK;
    });
}

// ---- the synthetic dense-Hessian NLP (acopf_dev.hpp dense_eval): structure as sqpsolver.jl_amd/dense_synth.py lays it out
extern "C" int sqphip_dense_attach(sqphip_ctx *h, const double *Q, const double *A, double kappa)
{
    if (!h || !Q || !A || !(kappa >= 0.0)) return SQPHIP_EINVAL;
    Ctx &C0 = h->c;
    const long n = C0.d.n, m = C0.d.m;
    if (C0.d.nnzj_coo != m * n || C0.d.nnzh_coo != n * (n + 1) / 2 || C0.d.nlin != m) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) {
        DV &d = C.d;
        double *q = C.dalloc<double>((size_t)(n * n)), *a = C.dalloc<double>((size_t)(m * n));
        h2d(C, q, Q, (size_t)(n * n)); h2d(C, a, A, (size_t)(m * n));
        d.dnQ = q; d.dnA = a; d.dn_kappa = kappa;
        d.dnc = C.dalloc<double>((size_t)d.B * n);
        d.dense_nlp = 1;
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        C.acopf_attached = true;           // (the batched run! has its device callbacks)
        make_lanes(C);
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_dense_set_instance(sqphip_ctx *h, int32_t inst, const double *c, const double *x0)
{
    if (!h || !h->c.d.dense_nlp || inst < 0 || inst >= h->c.d.B || !c || !x0) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) {
        DV &d = C.d;
        h2d(C, d.dnc + (size_t)inst * d.n, c, d.n);
        h2d(C, d.x0 + (size_t)inst * d.n, x0, d.n);
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_acopf_eval(sqphip_ctx *h, int32_t inst, const double *x, double sigma, const double *lambda,
                                 double *f, double *grad, double *g, double *jval, double *hval)
{
    if (!h || !h->c.acopf_attached || inst < 0 || inst >= h->c.d.B || !x) return SQPHIP_EINVAL;
    if (hval && !lambda) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) {
        DV &d = C.d;
        // staging in instance-`inst` scratch vectors
        double *xd = d.tmpx + (size_t)inst * d.n, *ld = d.hlam + (size_t)inst * d.m;
        double *gd = d.tmpE + (size_t)inst * d.m, *grd = d.wn + (size_t)inst * d.n;
        double *jd = d.jcoo + (size_t)inst * d.nnzj_coo, *hd = d.hcoo + (size_t)inst * d.nnzh_coo;
        double *fd = d.wN + (size_t)inst * d.Npad;
        h2d(C, xd, x, d.n);
        if (lambda) h2d(C, ld, lambda, d.m);
        launch_acopf_eval_point(C, inst, xd, sigma, lambda ? ld : nullptr, f ? fd : nullptr, grad ? grd : nullptr,
                                g ? gd : nullptr, jval ? jd : nullptr, hval ? hd : nullptr);
        d2h(C, f, fd, 1); d2h(C, grad, grd, d.n); d2h(C, g, gd, d.m);
        d2h(C, jval, jd, d.nnzj_coo); d2h(C, hval, hd, d.nnzh_coo);
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_sqp_reset(sqphip_ctx *h)
{
    if (!h || !h->c.acopf_attached) return SQPHIP_ESTATE;
    return guarded(h, [&](Ctx &C) { sqp_reset(C); return SQPHIP_OK; });
}

extern "C" int sqphip_sqp_run(sqphip_ctx *h, int32_t max_outer)
{
    if (!h || !h->c.acopf_attached) return SQPHIP_ESTATE;
    return guarded(h, [&](Ctx &C) {
        auto t0 = std::chrono::steady_clock::now();
        sqp_run(C, max_outer);
        C.total_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_sqp_get(sqphip_ctx *h, int32_t inst, double *x, double *g, double *mult_g, double *mult_x_L,
                              double *mult_x_U, double *obj_val, int32_t *status, int32_t *iter)
{
    if (!h || inst < 0 || inst >= h->c.d.B) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) {
        DV &d = C.d;
        SqpState S;
        SQPHIP_HIP_OK(hipMemcpyAsync(&S, d.sst + inst, sizeof(SqpState), hipMemcpyDeviceToHost, C.stream));
        d2h(C, x, d.x + (size_t)inst * d.n, d.n); d2h(C, g, d.E + (size_t)inst * d.m, d.m);
        d2h(C, mult_g, d.lambda + (size_t)inst * d.m, d.m);
        d2h(C, mult_x_L, d.mxL + (size_t)inst * d.n, d.n); d2h(C, mult_x_U, d.mxU + (size_t)inst * d.n, d.n);
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        // sqp_trust_region.jl:219-220: mult_g = -lambda, mult_x_U = -mult_x_U
        if (mult_g) for (int i = 0; i < d.m; ++i) mult_g[i] = -mult_g[i];
        if (mult_x_U) for (int j = 0; j < d.n; ++j) mult_x_U[j] = -mult_x_U[j];
        if (obj_val) *obj_val = S.obj_val;
        if (status) *status = S.ret;
        if (iter) *iter = S.iter;
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_sqp_status(sqphip_ctx *h, int32_t *ret_codes, int32_t *iters, int32_t *done)
{
    if (!h) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) {
        std::vector<SqpState> S(C.d.B);
        SQPHIP_HIP_OK(hipMemcpyAsync(S.data(), C.d.sst, sizeof(SqpState) * C.d.B, hipMemcpyDeviceToHost, C.stream));
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        for (int b = 0; b < C.d.B; ++b) {
            if (ret_codes) ret_codes[b] = S[b].ret;
            if (iters) iters[b] = S[b].iter;
            if (done) done[b] = S[b].done;
        }
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_sqp_trace(sqphip_ctx *h, int32_t inst, double *rows, int32_t cap, int32_t *len)
{
    if (!h || inst < 0 || inst >= h->c.d.B || !len) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) {
        SqpState S;
        SQPHIP_HIP_OK(hipMemcpyAsync(&S, C.d.sst + inst, sizeof(SqpState), hipMemcpyDeviceToHost, C.stream));
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        *len = S.trace_len;
        const int k = std::min<int>({cap, S.trace_len, SQPHIP_TRACE_CAP});
        if (rows && k > 0)
            d2h(C, rows, C.d.trace + (size_t)inst * SQPHIP_TRACE_CAP * SQPHIP_TRACE_COLS,
                (size_t)k * SQPHIP_TRACE_COLS);
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_get_counters(sqphip_ctx *h, sqphip_counters *c)
{
    if (!h || !c) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) {
        std::vector<SqpState> S(C.d.B);
        SQPHIP_HIP_OK(hipMemcpyAsync(S.data(), C.d.sst, sizeof(SqpState) * C.d.B, hipMemcpyDeviceToHost, C.stream));
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        C.tm.flush();
        double lane_factor = 0, lane_trailing = 0, lane_solve = 0;
        long lane_sweeps = 0;
        for (auto &L : C.lanes) {
            L->tm.flush();
            lane_factor += L->tm.factor_seconds; lane_trailing += L->tm.trailing_seconds; lane_solve += L->tm.solve_seconds;
            lane_sweeps += L->n_sweeps;
        }
        int64_t nqp = C.n_qp, nip = C.n_ipm_iter, nf = C.n_factor, nsol = C.n_solve;
        for (auto &s : S) { nqp += s.n_qp; nip += s.tot_ipm; nf += s.tot_fac; nsol += s.tot_sol; }
        c->n_qp = nqp; c->n_ipm_iter = nip; c->n_factor = nf;
        const double N = (double)C.d.Nf;
        c->kkt_order = C.d.Nf;
        c->lead_tiles = C.d.Ts;
        c->trailing_flops_per_factor = ldlt_trailing_flops(C.plan);
        c->ldlt_flops = (double)nf * N * N * N / 3.0;
        c->ldlt_seconds = C.tm.factor_seconds + lane_factor; c->trailing_seconds = C.tm.trailing_seconds + lane_trailing;
        c->solve_seconds = C.tm.solve_seconds + lane_solve; c->total_seconds = C.total_seconds;
        c->trailing_launches = C.tm.trailing_launches;
        c->n_sweeps = C.n_sweeps + lane_sweeps; c->n_solve = nsol; c->n_groups = C.lanes.empty() ? 1 : (int64_t)C.lanes.size();
        c->sparse = C.d.sparse; c->nnz_k = 0; c->nnz_l = 0; c->n_supernodes = 0; c->n_levels = 0; c->max_front = 0;
        c->factor_flops = 0; c->front_doubles = 0; c->cb_doubles = 0; c->factor_launches = 0; c->solve_launches = 0;
        c->nnz_l_top = 0; c->nnz_k_top = 0; c->cols_top = 0;
        if (C.d.sparse) {
            const SparseSym &Y = C.mfp().S;
            c->nnz_k = C.mfp().nnzK; c->nnz_l = Y.nnzL; c->n_supernodes = Y.ns; c->n_levels = Y.nlevels;
            c->max_front = Y.max_front; c->factor_flops = Y.flops; c->front_doubles = C.mfp().stride;
            long cb = 0;
            for (int q = 0; q < Y.ns; ++q) cb += (long)(Y.sn_nr[q] + 1) * Y.sn_nr[q] - (long)Y.sn_nr[q] * (Y.sn_nr[q] - 1) / 2;
            c->cb_doubles = cb;
            long lt = 0, kt = 0, nct = 0;
            for (int q = 0; q < Y.ns; ++q)
                if (Y.sn_level[q] >= C.mfp().narrow_level) {
                    const long nc = Y.sn_nc[q], nr = Y.sn_nr[q];
                    lt += nc * (nc - 1) / 2 + nc * nr; kt += C.mfp().asm_ptr[q + 1] - C.mfp().asm_ptr[q]; nct += nc;
                }
            c->nnz_l_top = lt; c->nnz_k_top = kt; c->cols_top = nct;
            c->factor_launches = (int64_t)(C.d.mf.sp_n > 0 ? C.mfp().fac_below + 1 : (int)C.mfp().fac.size());   // (+ k_mf_values)
            c->solve_launches = (int64_t)(C.mfp().fwd.size() + C.mfp().bwd.size() + (C.mfp().top.count > 0 ? 1 : 0));
            c->ldlt_flops = (double)nf * Y.flops;
        }
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_get_mode_counters(sqphip_ctx *h, int64_t *out)
{
    if (!h || !out) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) -> int {
        for (int k = 0; k < 12; ++k) out[k] = 0;
        if (!C.d.sst) return SQPHIP_OK;
        std::vector<SqpState> S((size_t)C.d.B);
        SQPHIP_HIP_OK(hipMemcpyAsync(S.data(), C.d.sst, sizeof(SqpState) * C.d.B, hipMemcpyDeviceToHost, C.stream));
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        for (auto &s : S)
            for (int k = 0; k < 4; ++k) { out[3 * k] += s.md_qp[k]; out[3 * k + 1] += s.md_ipm[k]; out[3 * k + 2] += s.md_fac[k]; }
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_sqp_work(sqphip_ctx *h, int64_t *qp, int64_t *ipm, int64_t *fac)
{
    if (!h) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) -> int {
        if (!C.d.sst) return SQPHIP_EINVAL;
        std::vector<SqpState> S((size_t)C.d.B);
        SQPHIP_HIP_OK(hipMemcpyAsync(S.data(), C.d.sst, sizeof(SqpState) * C.d.B, hipMemcpyDeviceToHost, C.stream));
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        for (int b = 0; b < C.d.B; ++b) {
            if (qp) qp[b] = S[b].n_qp;
            if (ipm) ipm[b] = S[b].tot_ipm;
            if (fac) fac[b] = S[b].tot_fac;
        }
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_sqp_qp_log(sqphip_ctx *h, int32_t inst, int32_t *rows, int32_t cap, int32_t *n_rows)
{
    if (!h || !rows || !n_rows || cap < 0) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) -> int {
        if (!C.d.sst || inst < 0 || inst >= C.d.B) return SQPHIP_EINVAL;
        SqpState S;
        SQPHIP_HIP_OK(hipMemcpyAsync(&S, C.d.sst + inst, sizeof(SqpState), hipMemcpyDeviceToHost, C.stream));
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        const int have = S.qlog_n < SQPHIP_QLOG_CAP ? S.qlog_n : SQPHIP_QLOG_CAP, n = have < cap ? have : cap;
        for (int k = 0; k < n; ++k) {
            const int src = (S.qlog_n - n + k) % SQPHIP_QLOG_CAP;
            for (int c = 0; c < 4; ++c) rows[4 * k + c] = S.qlog[4 * src + c];
        }
        *n_rows = n;
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_sqp_qp_log_term(sqphip_ctx *h, int32_t inst, double *scaled_error, int32_t *rule, int32_t cap, int32_t *n_rows)
{
    if (!h || !n_rows || cap < 0) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) -> int {
        if (!C.d.sst || inst < 0 || inst >= C.d.B) return SQPHIP_EINVAL;
        SqpState S;
        SQPHIP_HIP_OK(hipMemcpyAsync(&S, C.d.sst + inst, sizeof(SqpState), hipMemcpyDeviceToHost, C.stream));
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        const int have = S.qlog_n < SQPHIP_QLOG_CAP ? S.qlog_n : SQPHIP_QLOG_CAP, n = have < cap ? have : cap;
        for (int k = 0; k < n; ++k) {
            const int src = (S.qlog_n - n + k) % SQPHIP_QLOG_CAP;
            if (scaled_error) scaled_error[k] = (double)S.qerr[src];
            if (rule) rule[k] = (int32_t)S.qrule[src];
        }
        *n_rows = n;
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_get_termination_counters(sqphip_ctx *h, int64_t *out)
{
    if (!h || !out) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) -> int {
        for (int k = 0; k < 4; ++k) out[k] = 0;
        if (!C.d.sst) return SQPHIP_OK;
        std::vector<SqpState> S((size_t)C.d.B);
        SQPHIP_HIP_OK(hipMemcpyAsync(S.data(), C.d.sst, sizeof(SqpState) * C.d.B, hipMemcpyDeviceToHost, C.stream));
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        for (auto &s : S) for (int k = 0; k < 4; ++k) out[k] += s.term_rule[k];
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_sqp_last_request(sqphip_ctx *h, int32_t inst, int32_t *mode, double *delta, double *mu_pen,
                                       double *x_k, double *c, double *b, double *jac_coo, double *hess_coo)
{
    if (!h) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) -> int {
        const DV &d = C.d;
        if (!d.ist || inst < 0 || inst >= d.B) return SQPHIP_EINVAL;
        IpmState I;
        SQPHIP_HIP_OK(hipMemcpyAsync(&I, d.ist + inst, sizeof(IpmState), hipMemcpyDeviceToHost, C.stream));
        auto get = [&](double *dst, const double *src, long stride) {
            if (dst && stride > 0)
                SQPHIP_HIP_OK(hipMemcpyAsync(dst, src + inst * stride, sizeof(double) * stride, hipMemcpyDeviceToHost, C.stream));
        };
        get(x_k, d.xk, d.n); get(c, d.cin, d.n); get(b, d.bE, d.m);
        get(jac_coo, d.jcoo, d.nnzj_coo); get(hess_coo, d.hcoo, d.nnzh_coo);
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        if (mode) *mode = I.mode;
        if (delta) *delta = I.delta;
        if (mu_pen) *mu_pen = I.mu_pen;
        return SQPHIP_OK;
    });
}

// ---- scenario queue: more scenarios than slots ----------------------------------------------------------------
extern "C" int sqphip_sqp_stream_begin(sqphip_ctx *h, int32_t n_scenarios)
{
    if (!h || !h->c.acopf_attached || n_scenarios <= 0) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) {
        DV &d = C.d;
        StreamDev &Q = d.stream;
        const size_t M = (size_t)n_scenarios;
        Q.M = n_scenarios;
        Q.next = C.dalloc<int>(1); Q.slot_scen = C.dalloc<int>((size_t)d.B);
        Q.qend = C.dalloc<int>(1);
        {   // hand-out order: identity, all M scenarios (sqphip_sqp_stream_assign changes it)
            std::vector<int> ids(M);
            for (size_t k = 0; k < M; ++k) ids[k] = (int)k;
            Q.qids = C.upload(ids);
            SQPHIP_HIP_OK(hipMemcpyAsync(Q.qend, &Q.M, sizeof(int), hipMemcpyHostToDevice, C.stream));
        }
        C.stream_started = false;
        Q.xL = C.dalloc<double>(M * d.n); Q.xU = C.dalloc<double>(M * d.n); Q.x0 = C.dalloc<double>(M * d.n);
        Q.gL = C.dalloc<double>(M * d.m); Q.gU = C.dalloc<double>(M * d.m);
        Q.ohm = C.dalloc<double>(M * d.nl * 12); Q.c2 = C.dalloc<double>(M * d.ng); Q.c1 = C.dalloc<double>(M * d.ng);
        Q.rx = C.dalloc<double>(M * d.n); Q.robj = C.dalloc<double>(M);
        Q.rstat = C.dalloc<int>(M); Q.riter = C.dalloc<int>(M);
        // every slot starts as "queue exhausted" (-1): the stage kernel's queue step stays off until
        // sqphip_sqp_stream_run arms the slots (-2), so a plain sqp_reset / sqp_run between _begin and _stream_run
        // behaves as if no queue existed (dalloc zero-fills, and 0 is a valid scenario id)
        SQPHIP_HIP_OK(hipMemsetAsync(Q.slot_scen, 0xff, sizeof(int) * (size_t)d.B, C.stream));
        // riter = -1: "no result filed by this context" (sqphip_sqp_stream_get; reset by _stream_run and _stream_assign)
        SQPHIP_HIP_OK(hipMemsetAsync(Q.riter, 0xff, sizeof(int) * M, C.stream));
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        make_lanes(C);
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_sqp_stream_set(sqphip_ctx *h, int32_t scen, const double *xL, const double *xU, const double *gL,
                                     const double *gU, const double *ohm, const double *c2, const double *c1,
                                     const double *x0)
{
    if (!h || scen < 0 || scen >= h->c.d.stream.M || !xL || !xU || !gL || !gU || !ohm || !c2 || !c1 || !x0) return SQPHIP_EINVAL;
    for (int i = 0; i < h->c.d.m; ++i) {
        if (gL[i] == -INFINITY && gU[i] == INFINITY) return SQPHIP_EINVAL;
        if (h->c.d.condense && gL[i] == gU[i] && h->c.h_kpos[i] < 0) {
            h->c.err = "sqphip_sqp_stream_set: row " + std::to_string(i) + " is an equality for this scenario but was not one when "
                       "the context was created (options.kkt_condense = 1 fixes the kept rows)";
            return SQPHIP_EINVAL;
        }
    }
    return guarded(h, [&](Ctx &C) {
        DV &d = C.d;
        const StreamDev &Q = d.stream;
        const size_t s = (size_t)scen;
        h2d(C, const_cast<double *>(Q.xL) + s * d.n, xL, d.n); h2d(C, const_cast<double *>(Q.xU) + s * d.n, xU, d.n);
        h2d(C, const_cast<double *>(Q.x0) + s * d.n, x0, d.n);
        h2d(C, const_cast<double *>(Q.gL) + s * d.m, gL, d.m); h2d(C, const_cast<double *>(Q.gU) + s * d.m, gU, d.m);
        h2d(C, const_cast<double *>(Q.ohm) + s * d.nl * 12, ohm, (size_t)d.nl * 12);
        h2d(C, const_cast<double *>(Q.c2) + s * d.ng, c2, d.ng); h2d(C, const_cast<double *>(Q.c1) + s * d.ng, c1, d.ng);
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_sqp_stream_run(sqphip_ctx *h)
{
    if (!h || !h->c.acopf_attached || h->c.d.stream.M <= 0) return SQPHIP_ESTATE;
    return guarded(h, [&](Ctx &C) {
        auto t0 = std::chrono::steady_clock::now();
        SQPHIP_HIP_OK(hipMemsetAsync(C.d.stream.next, 0, sizeof(int), C.stream));
        SQPHIP_HIP_OK(hipMemsetAsync(C.d.stream.riter, 0xff, sizeof(int) * (size_t)C.d.stream.M, C.stream));   // results of an earlier pass are not this pass's
        sqp_stream_arm(C);
        C.stream_started = true;
        sqp_run(C, 0);
        C.total_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return SQPHIP_OK;
    });
}

// ---- a queue shared between ranks (SURVEY.md section 8f-4 remainder): every rank holds the tables of ALL scenarios
// (sqphip_sqp_stream_set) but hands out only the ids assigned to it; between budgeted runs the host moves unstarted ids
// from a rank that still has many to one whose queue ran dry (sqpsolver.jl_amd/shard.py, run_shared_queue).  No iterate
// crosses ranks: a scenario is solved from start to end by the rank that drew it, and its result does not depend on
// which rank or slot that was.
static int stream_state(Ctx &C, int &next, int &end)
{
    SQPHIP_HIP_OK(hipMemcpyAsync(&next, C.d.stream.next, sizeof(int), hipMemcpyDeviceToHost, C.stream));
    SQPHIP_HIP_OK(hipMemcpyAsync(&end, C.d.stream.qend, sizeof(int), hipMemcpyDeviceToHost, C.stream));
    SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
    if (next > end) next = end;                     // slots that found the queue empty overshoot by one each
    return SQPHIP_OK;
}

extern "C" int sqphip_sqp_stream_assign(sqphip_ctx *h, int32_t n, const int32_t *ids)
{
    if (!h || h->c.d.stream.M <= 0 || n < 0 || n > h->c.d.stream.M || (n > 0 && !ids)) return SQPHIP_EINVAL;
    for (int k = 0; k < n; ++k) if (ids[k] < 0 || ids[k] >= h->c.d.stream.M) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) {
        const int zero = 0;
        if (n > 0) SQPHIP_HIP_OK(hipMemcpyAsync(C.d.stream.qids, ids, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, C.stream));
        SQPHIP_HIP_OK(hipMemcpyAsync(C.d.stream.qend, &n, sizeof(int), hipMemcpyHostToDevice, C.stream));
        SQPHIP_HIP_OK(hipMemcpyAsync(C.d.stream.next, &zero, sizeof(int), hipMemcpyHostToDevice, C.stream));
        SQPHIP_HIP_OK(hipMemsetAsync(C.d.stream.riter, 0xff, sizeof(int) * (size_t)C.d.stream.M, C.stream));   // nothing filed yet under the new assignment
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        C.stream_started = false;
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_sqp_stream_append(sqphip_ctx *h, int32_t n, const int32_t *ids)
{
    if (!h || h->c.d.stream.M <= 0 || n < 0 || (n > 0 && !ids)) return SQPHIP_EINVAL;
    for (int k = 0; k < n; ++k) if (ids[k] < 0 || ids[k] >= h->c.d.stream.M) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) {
        int next, end;
        stream_state(C, next, end);
        if (end + n > C.d.stream.M) { C.err = "sqphip_sqp_stream_append: more ids than scenario tables"; return SQPHIP_EINVAL; }
        if (n > 0) SQPHIP_HIP_OK(hipMemcpyAsync(C.d.stream.qids + end, ids, sizeof(int) * (size_t)n, hipMemcpyHostToDevice, C.stream));
        const int ne = end + n;
        SQPHIP_HIP_OK(hipMemcpyAsync(C.d.stream.qend, &ne, sizeof(int), hipMemcpyHostToDevice, C.stream));
        SQPHIP_HIP_OK(hipMemcpyAsync(C.d.stream.next, &next, sizeof(int), hipMemcpyHostToDevice, C.stream));   // undo the overshoot
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_sqp_stream_release(sqphip_ctx *h, int32_t n, int32_t *ids_out, int32_t *n_out)
{
    if (!h || h->c.d.stream.M <= 0 || n < 0 || !n_out || (n > 0 && !ids_out)) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) {
        int next, end;
        stream_state(C, next, end);
        const int k = std::min(n, end - next), ne = end - k;
        if (k > 0) SQPHIP_HIP_OK(hipMemcpyAsync(ids_out, C.d.stream.qids + ne, sizeof(int) * (size_t)k, hipMemcpyDeviceToHost, C.stream));
        SQPHIP_HIP_OK(hipMemcpyAsync(C.d.stream.qend, &ne, sizeof(int), hipMemcpyHostToDevice, C.stream));
        SQPHIP_HIP_OK(hipMemcpyAsync(C.d.stream.next, &next, sizeof(int), hipMemcpyHostToDevice, C.stream));
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        *n_out = k;
        return SQPHIP_OK;
    });
}

// every slot performs up to `max_outer` more outer iterations (of whatever scenarios it holds or draws); afterwards
// n_unstarted = ids of this rank's queue nobody has drawn yet, n_active = slots with a run in progress
extern "C" int sqphip_sqp_stream_run_some(sqphip_ctx *h, int32_t max_outer, int32_t *n_unstarted, int32_t *n_active)
{
    if (!h || !h->c.acopf_attached || h->c.d.stream.M <= 0 || max_outer < 1) return SQPHIP_ESTATE;
    return guarded(h, [&](Ctx &C) {
        auto t0 = std::chrono::steady_clock::now();
        if (!C.stream_started) { sqp_stream_arm(C); C.stream_started = true; }
        else sqp_stream_rearm(C);
        sqp_run(C, max_outer);
        C.total_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        int next, end;
        stream_state(C, next, end);
        if (n_unstarted) *n_unstarted = end - next;
        if (n_active) {
            std::vector<int> slot(C.d.B);
            std::vector<SqpState> S(C.d.B);
            SQPHIP_HIP_OK(hipMemcpyAsync(slot.data(), C.d.stream.slot_scen, sizeof(int) * C.d.B, hipMemcpyDeviceToHost, C.stream));
            SQPHIP_HIP_OK(hipMemcpyAsync(S.data(), C.d.sst, sizeof(SqpState) * C.d.B, hipMemcpyDeviceToHost, C.stream));
            SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
            int a = 0;
            for (int b = 0; b < C.d.B; ++b) a += slot[b] >= 0 && !S[b].done;
            *n_active = a;
        }
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_sqp_stream_get(sqphip_ctx *h, int32_t scen, double *x, double *obj_val, int32_t *status, int32_t *iter)
{
    if (!h || scen < 0 || scen >= h->c.d.stream.M) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) {
        const DV &d = C.d;
        const StreamDev &Q = d.stream;
        d2h(C, x, Q.rx + (size_t)scen * d.n, d.n);
        d2h(C, obj_val, Q.robj + scen, 1);
        int st = 0, it = 0;
        SQPHIP_HIP_OK(hipMemcpyAsync(&st, Q.rstat + scen, sizeof(int), hipMemcpyDeviceToHost, C.stream));
        SQPHIP_HIP_OK(hipMemcpyAsync(&it, Q.riter + scen, sizeof(int), hipMemcpyDeviceToHost, C.stream));
        SQPHIP_HIP_OK(hipStreamSynchronize(C.stream));
        if (status) *status = st;
        if (iter) *iter = it;
        return SQPHIP_OK;
    });
}

// seconds of kernel time by class (sqphip.h: sqphip_get_kernel_times), summed over the instance groups, and launch groups timed
extern "C" int sqphip_get_kernel_times(sqphip_ctx *h, double *seconds, int64_t *groups, int32_t cap)
{
    if (!h || !seconds || !groups || cap < 0) return SQPHIP_EINVAL;
    return guarded(h, [&](Ctx &C) -> int {
        C.tm.flush();
        for (auto &L : C.lanes) L->tm.flush();
        for (int c = 0; c < cap; ++c) {
            seconds[c] = 0.0; groups[c] = 0;
            if (c >= KC_COUNT) continue;
            seconds[c] = C.tm.class_seconds[c]; groups[c] = C.tm.class_groups[c];
            for (auto &L : C.lanes) { seconds[c] += L->tm.class_seconds[c]; groups[c] += L->tm.class_groups[c]; }
        }
        return SQPHIP_OK;
    });
}

extern "C" int sqphip_reset_counters(sqphip_ctx *h)
{
    if (!h) return SQPHIP_EINVAL;
    Ctx &C = h->c;
    C.tm.flush();
    C.tm.trailing_seconds = C.tm.factor_seconds = C.tm.solve_seconds = 0;
    C.tm.trailing_launches = 0; C.tm.n_factor = 0;
    for (int c = 0; c < KC_COUNT; ++c) { C.tm.class_seconds[c] = 0; C.tm.class_groups[c] = 0; }
    C.n_qp = C.n_ipm_iter = C.n_factor = C.n_solve = 0; C.total_seconds = 0; C.n_sweeps = 0;
    for (auto &L : C.lanes) {
        L->tm.flush();
        L->tm.trailing_seconds = L->tm.factor_seconds = L->tm.solve_seconds = 0; L->tm.trailing_launches = 0; L->tm.n_factor = 0;
        for (int c = 0; c < KC_COUNT; ++c) { L->tm.class_seconds[c] = 0; L->tm.class_groups[c] = 0; }
        L->n_sweeps = 0;
    }
    return SQPHIP_OK;
}

// enable / disable HIP-event timing of the factor, trailing-update and solve kernels
extern "C" int sqphip_set_timing(sqphip_ctx *h, int32_t enabled)
{
    if (!h) return SQPHIP_EINVAL;
    h->c.tm.flush();
    h->c.tm.enabled = enabled != 0; h->c.tm.detail = enabled == 2;
    for (auto &L : h->c.lanes) { L->tm.flush(); L->tm.enabled = enabled != 0; L->tm.detail = enabled == 2; }
    return SQPHIP_OK;
}
