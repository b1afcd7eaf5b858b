// mfront.hip -- batched multifrontal LDL^T (no pivoting) of the sparse Newton matrix and its triangular solves,
// for gfx950 (MI355X).
//
// This is the arithmetic the reference leaves to Ipopt's sparse linear solver (MUMPS / MA57,
// /root/reference/examples/acopf/opf.jl:59-64) behind JuMP.optimize! (/root/reference/src/algorithms/
// subproblem_JuMP.jl:178): numeric factorisation of the interior-point Newton matrix of every QP sub-problem, once
// per interior-point iteration, and 2-4 solves with the factors.
//
// Plan (mfplan.hip / symbolic.hip, built once per sparsity structure): supernodes in postorder, one dense front per
// supernode in the instance's front arena, fronts grouped by level of the assembly tree.  Numeric phase, level by
// level from the leaves:
//   k_mf_factor   one workgroup per (front, instance): zero the front (in LDS for fronts up to 90 rows), gather the
//                 structural entries of the Newton matrix from the sub-problem data (item lists: every destination
//                 summed in a fixed order), extend-add the contribution blocks of the children (one child after the
//                 other), eliminate the supernode's columns (right-looking rank-1 updates, one barrier per column),
//                 write L and the contribution block back.  An extra last ROW of the front carries the right-hand
//                 side through the same elimination: the forward solve of the first right-hand side costs nothing.
//   k_mf_fwd      further right-hand sides: per front y = b_cols + sum of the children's updates, triangular solve
//                 with L11, update vector for the ancestors (stored where the contribution block's last row lives).
//   k_mf_bwd      levels in reverse: x_cols = L11^-T (D^-1 y - L21' x_rows), x_rows gathered from the finished ancestors.
// No atomics anywhere: every entry has one writer per launch and every sum a fixed order, so a batch gives the same
// bits on every run.  HBM traffic per factorisation and instance: the item data (12 B per structural entry),
// 8 nnz(L) written, the contribution blocks written once and read once.
#include "ctx.hpp"
#include "dev_util.hpp"
#include "mf_dev.hpp"
#include <cstdlib>
#include <mutex>

namespace sqphip {

// phase stamps of the factor kernel (instance 0 only; scripts/gpu_mf_trace.py builds with -DSQPHIP_MF_TRACE)
#ifdef SQPHIP_MF_TRACE
__device__ long long g_mf_trace[1 << 16][8];
#define MF_TR(i) if (inst == 0 && threadIdx.x == 0 && s < (1 << 16)) g_mf_trace[s][i] = (long long)wall_clock64();
// shader-clock stamps inside the static front kernel (wave 0 of instance 0): [front][16]
__device__ long long g_mf_trace2[1 << 12][16];
#define MF_TRW(i) if (W == 0 && trs >= 0 && trs < (1 << 12) && lane == 0) g_mf_trace2[trs][i] = (long long)clock64();
// ... and inside the four-wave solve routines (thread 0 of instance 0): [front][forward 0..7 | backward 8..15]
__device__ long long g_mf_trace3[1 << 12][16];
#define MF_TRS(i) if (inst == 0 && tid == 0 && s < (1 << 12)) g_mf_trace3[s][i] = (long long)clock64();
// ... and in the streamed top-of-tree solve: per step [compute wave: start, done, past the barrier | -, loader wave 1: likewise]
#define MF_TR2S(a, b) if (inst == 0 && lane == 0 && j < 64 && wave < 2) g_mf_trace3[(do_fwd ? 64 : 0) + j][wave == 0 ? (a) : (b)] = (long long)clock64();
#define MF_TR2W(c) if (trj >= 0 && lane == 0) g_mf_trace3[trj][c] = (long long)clock64();
// ... and in the spine kernel (thread 0 of instance 0, first candidate): rows 128 + k, 100 MHz wall clock
#define MF_TRSP(c) if (inst == 0 && cand == 0 && tid == 0 && k < 64) g_mf_trace3[128 + k][c] = (long long)wall_clock64();
#else
#define MF_TR(i)
#define MF_TRW(i)
#define MF_TRS(i)
#define MF_TR2S(a, b)
#define MF_TR2W(c)
#define MF_TRSP(c)
#endif

// Thread layout inside a front: RL row lanes x (NT / RL) column groups; RL = 16 / 32 / 64 by front height so that a
// tiny front does not idle three quarters of its lanes.
template <int NT, bool INPLACE>
__global__ __launch_bounds__(NT) void k_mf_factor(DV d, int sbegin, int want, int with_rhs)
{
    int inst, cand;
    if (!mf_candidate(d, want, inst, cand)) return;
    const MfDev &M = d.mf;
    const int s = M.sched[sbegin + blockIdx.x];
    const MfFrontDesc Fd = M.desc[s];
    const int nc = Fd.nc, nr = Fd.nr, fs = nc + nr, ld = fs + 1, f0 = Fd.first;
    double *G = mf_arena(d, inst, cand) + Fd.off;
    extern __shared__ double mf_lds[];
    double *F = INPLACE ? G : mf_lds;
    const int tid = threadIdx.x;
    const int rl_log = ld <= 16 ? 4 : (ld <= 32 ? 5 : 6);
    const int RL = 1 << rl_log, rlane = tid & (RL - 1), cg = tid >> rl_log, ncg = NT >> rl_log;
    MF_TR(0)
    // 1. zero
    for (int e = tid; e < ld * fs; e += NT) F[e] = 0.0;
    __syncthreads();
    MF_TR(1)
    // 2. structural entries of the Newton matrix that live in this front, right-hand side row
    {
        const IpmState &st = d.ist[inst];
        const double hsc = st.hsc, dw = st.dw;
        const double *hv = d.hv + (long)inst * d.nnzhc, *jv = d.jv + (long)inst * d.nnzjc;
        const double *Dd = d.Dd + (long)inst * d.m, *sigp = d.sigp + (long)inst * d.n, *hd = d.hd + (long)inst * d.n;
        const int *rt = d.rtype + (long)inst * d.m;
        for (int e = Fd.asm_begin + tid; e < Fd.asm_end; e += NT) {
            double a = 0.0;
            for (int k = M.item_ptr[e]; k < M.item_ptr[e + 1]; ++k) a += mf_item_value(M.items[k], hv, jv, Dd, sigp, hd, rt, hsc, dw);
            F[M.dest_loc[e]] = a;
        }
        if (with_rhs) {
            const double *b = d.xv + (long)inst * d.Fpad + f0;
            for (int j = tid; j < nc; j += NT) F[j * ld + fs] = b[j];
        }
    }
    __syncthreads();
    MF_TR(2)
    // 3. extend-add of the children's contribution blocks (rows cnc.. of columns cnc.. of the child's front, its
    //    right-hand-side row included), one child after the other: two children may hit the same entry
    for (int q = M.child_ptr[s]; q < M.child_ptr[s + 1]; ++q) {
        const int c = M.child[q];
        const int cnc = M.nc[c], cnr = M.nr[c], cld = cnc + cnr + 1;
        const double *Cg = mf_arena(d, inst, cand) + M.off[c] + (long)cnc * cld + cnc;
        const int *rel = M.rel + M.rowptr[c];
        for (int jj = cg; jj < cnr; jj += ncg) {
            const int gj = rel[jj] * ld;
            const double *Cc = Cg + (long)jj * cld;
            for (int ii = jj + rlane; ii <= cnr; ii += RL) F[gj + (ii < cnr ? rel[ii] : fs)] += Cc[ii];
        }
        __syncthreads();
    }
    MF_TR(3)
    // 4. eliminate the nc columns of the supernode: right-looking, one barrier per column.  Column k keeps the
    //    unscaled entries d_k L_ik (nobody writes it after step k - 1); L_jk = F_jk / d_k is formed on the fly.
    for (int k = 0; k < nc; ++k) {
        const double *Fk = F + k * ld;
        const double di = 1.0 / Fk[k];
        for (int j = k + 1 + cg; j < fs; j += ncg) {
            const double lj = Fk[j] * di;
            double *Fj = F + j * ld;
            for (int i = j + rlane; i <= fs; i += RL) Fj[i] -= Fk[i] * lj;
        }
        __syncthreads();
    }
    MF_TR(4)
    // 5. results: L (scaled) and 1 / D, z = D^-1 L^-1 b, contribution block with its right-hand-side row
    double *dinv = mf_dinv(d, inst, cand) + f0, *vv = mf_vv(d, inst, cand) + f0;
    for (int j = cg; j < fs; j += ncg) {
        const double *Fj = F + j * ld;
        double *Gj = G + (long)j * ld;
        if (j < nc) {
            const double di = 1.0 / Fj[j];
            for (int i = j + 1 + rlane; i < fs; i += RL) Gj[i] = Fj[i] * di;
            if (rlane == 0) { dinv[j] = di; if (with_rhs) vv[j] = Fj[fs] * di; }
        } else if (!INPLACE) {
            for (int i = j + rlane; i <= fs; i += RL) Gj[i] = Fj[i];
        }
    }
    MF_TR(5)
}

#ifdef SQPHIP_MF_TRACE
extern "C" int sqphip_mf_trace_read(long long *out, int nfronts)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mf_trace), sizeof(long long) * 8 * (size_t)nfronts);
}
extern "C" int sqphip_mf_trace3_read(long long *out, int nfronts)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mf_trace3), sizeof(long long) * 16 * (size_t)nfronts);
}
extern "C" int sqphip_mf_trace2_read(long long *out, int nfronts)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mf_trace2), sizeof(long long) * 16 * (size_t)nfronts);
}
#endif

// ---------------------------------------------------------------------------------------------------------------
// values of the structural entries of the Newton matrix: one thread per destination, the items of a destination
// summed in list order.  Flat over the whole matrix and batch, so the latency-bound gather (item -> slot -> value)
// runs at full occupancy and off the level-by-level critical path of the front kernels.
__global__ __launch_bounds__(256) void k_mf_values(DV d, int want)
{
    int inst, cand;
    if (!mf_candidate(d, want, inst, cand)) return;
    const MfDev &M = d.mf;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= M.nnzK) return;
    const IpmState &st = d.ist[inst];
    const double dw = cand ? next_shift(st.dw, st.dw_last) : st.dw;
    const double *hv = d.hv + (long)inst * d.nnzhc, *jv = d.jv + (long)inst * d.nnzjc;
    const double *Dd = d.Dd + (long)inst * d.m, *sigp = d.sigp + (long)inst * d.n, *hd = d.hd + (long)inst * d.n;
    const int *rt = d.rtype + (long)inst * d.m;
    double a = 0.0;
    for (int k = M.item_ptr[e]; k < M.item_ptr[e + 1]; ++k) a += mf_item_value(M.items[k], hv, jv, Dd, sigp, hd, rt, st.hsc, dw);
    mf_vals(d, inst, cand)[e] = a;
}

// ---------------------------------------------------------------------------------------------------------------
// The front kernel: one workgroup of NW waves per (front, instance).
//   1. front image (LDS for fronts up to 80 rows, else the front's own storage in the arena): zero, the assembled
//      values of its structural entries, the right-hand side row, the children's contribution blocks through the
//      gather lists of the plan (one thread per receiving entry, sources in fixed order -- no barrier per child);
//   2. the image moves into MFMA accumulator registers: the lower-triangular 16 x 16 tiles of the front are dealt
//      round-robin to the waves, lane (l15, l4), element rr of a tile <-> row 16 ti + l15, column 16 tj + l4 + 4 rr;
//   3. elimination four columns at a time: the 4 x 4 diagonal block goes through LDS to every lane (a uniform
//      scalar LDL^T), each row of the panel is solved by three wave shuffles between the lane groups l4 = 0..3 that
//      hold its four entries, the panel (X = L D and L, rows below the block only) is published in LDS and every
//      tile receives the rank-4 update as ONE v_mfma_f64_16x16x4_f64: the accumulator layout of the panel is the
//      B-operand layout already.  Two barriers per four columns; nothing but the panel touches LDS.
//   4. L (scaled), 1 / D, D^-1 L^-1 b and the contribution block go back to the arena from the registers.
// Explicit zeros of the padding (rows / columns beyond the front) stay zero: an all-zero panel row updates nothing.
typedef double d4 __attribute__((ext_vector_type(4)));

// 1 / x by v_rcp_f64 and two Newton steps: a quarter of the dependent instructions of an IEEE division, within one ulp
// (four of these sit on the critical path of every four-column block).  0 gives inf -> NaN, which k_inertia reports.
#ifndef SQPHIP_MF_RCP_NR
#define SQPHIP_MF_RCP_NR 2
#endif
#ifndef SQPHIP_MF_LDL4_FAST
#define SQPHIP_MF_LDL4_FAST 0
#endif
__device__ __forceinline__ double mf_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
#if SQPHIP_MF_RCP_NR >= 2
    r = fma(fma(-x, r, 1.0), r, r);
#endif
    return r;
}

__device__ __forceinline__ double mf_readlane(double x, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane), hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// Row broadcasts of a wave seen as four rows of sixteen lanes (lane = l15 + 16 l4): r_k[lane] = x[l15 + 16 k].  gfx950 has two
// VALU lane swaps for this -- v_permlane16_swap (odd rows of one operand against the even rows of the other) and
// v_permlane32_swap (upper half against lower half) --: (x, x) -> (r0 r0 r2 r2), (r1 r1 r3 r3) -> (r0 r0 r0 r0), (r2 ..), (r1 ..):
// six VALU instructions for the three rows of a double where __shfl costs three ds_bpermute round trips through the LDS
// pipe (~120 cycles each; round 4, cycle stamps of the first 4 x 4 block of a diagonal tile: rows x inverse 364 -> see
// profiles/r04_front_diag_stamps.txt).  Same values, so the same bits.  SQPHIP_MF_SHFL=1 (build) keeps the shuffles.
#ifndef SQPHIP_MF_SHFL
#define SQPHIP_MF_SHFL 0
#endif
#ifndef SQPHIP_MF_EARLY_STORE
#define SQPHIP_MF_EARLY_STORE 1      // the level launches of the static front kernels store L step by step (mf_front_elim, EARLY), from five tile rows on
#endif
struct MfRows { double r0, r1, r2; };
__device__ __forceinline__ double mf_hilo(unsigned hi, unsigned lo) { return __hiloint2double((int)hi, (int)lo); }
__device__ __forceinline__ MfRows mf_rows3(double x, int l15)
{
#if SQPHIP_MF_SHFL
    return {__shfl(x, l15), __shfl(x, l15 + 16), __shfl(x, l15 + 32)};
#else
    (void)l15;
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto pl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false), ph = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const auto al = __builtin_amdgcn_permlane32_swap(pl[0], pl[0], false, false), ah = __builtin_amdgcn_permlane32_swap(ph[0], ph[0], false, false);
    const auto bl = __builtin_amdgcn_permlane32_swap(pl[1], pl[1], false, false), bh = __builtin_amdgcn_permlane32_swap(ph[1], ph[1], false, false);
    return {mf_hilo(ah[0], al[0]), mf_hilo(bh[0], bl[0]), mf_hilo(ah[1], al[1])};
#endif
}

template <int NW, int MAXT, bool LDSIMG>
__global__ __launch_bounds__(64 * NW) void k_mf_factor2(DV d, int sbegin, int want, int with_rhs, int Tl)
{
    constexpr int NT = 64 * NW;
    int inst, cand;
    if (!mf_candidate(d, want, inst, cand)) return;
    const MfDev &M = d.mf;
    const int s = M.sched[sbegin + blockIdx.x];
    const MfFrontDesc Fd = M.desc[s];
    const int nc = Fd.nc, nr = Fd.nr, fs = nc + nr, ld = fs + 1, f0 = Fd.first;
    const double *arena = mf_arena(d, inst, cand);
    double *G = mf_arena(d, inst, cand) + Fd.off;
    extern __shared__ double mf_lds[];
    const int R = 16 * Tl;
    double *Xp = mf_lds + (LDSIMG ? R * R : 0), *Lp = Xp + 4 * R, *blk = Lp + 4 * R, *dl = blk + 16;
    double *F = LDSIMG ? mf_lds : G;
    const int LD = LDSIMG ? R : ld;
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int T = (ld + 15) >> 4;
    MF_TR(0)
    // 1. image
    if (LDSIMG) {
        // the 16 T x 16 T corner of the R x R image; R = 16 T for most fronts of a launch: one flat loop
        if (T == Tl) for (int e = tid; e < R * R; e += NT) F[e] = 0.0;
        else for (int e = tid; e < 256 * T * T; e += NT) F[(e / (16 * T)) * LD + e % (16 * T)] = 0.0;
    } else for (int e = tid; e < ld * fs; e += NT) F[e] = 0.0;
    __syncthreads();
    MF_TR(1)
    {
        const double *vals = mf_vals(d, inst, cand);
        for (int e = Fd.asm_begin + tid; e < Fd.asm_end; e += NT) {
            const int rc = M.dest_rc[e];
            F[(rc >> 16) * LD + (rc & 0xffff)] = vals[e];
        }
        if (with_rhs) {
            const double *b = d.xv + (long)inst * d.Fpad + f0;
            for (int j = tid; j < nc; j += NT) F[j * LD + fs] = b[j];
        }
    }
    __syncthreads();
    MF_TR(2)
    for (int t = Fd.ea_begin + tid; t < Fd.ea_end; t += NT) {
        const MfGather g = M.ea_ent[t];
        double a = arena[g.src0];
        for (int q = g.src_begin + 1; q < g.src_end; ++q) a += arena[M.ea_src[q]];
        const int rc = g.where;
        F[(rc >> 16) * LD + (rc & 0xffff)] += a;
    }
    __syncthreads();
    MF_TR(3)
    // 2. tiles into registers
    int ti_[MAXT], tj_[MAXT];
    d4 acc[MAXT];
#pragma unroll
    for (int q = 0; q < MAXT; ++q) {
        int rem = q * NW + wave, tj = 0;
        while (tj < T && rem >= T - tj) { rem -= T - tj; ++tj; }
        const bool valid = tj < T;
        const int ti = tj + rem;
        ti_[q] = valid ? ti : -1; tj_[q] = tj;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int col = 16 * tj + l4 + 4 * rr, row = 16 * ti + l15;
            const bool in = valid && (LDSIMG || (row <= fs && col < fs));
            acc[q][rr] = in ? F[col * LD + row] : 0.0;
        }
    }
    __syncthreads();
    // 3. elimination, four columns at a time
    for (int tk = 0; 16 * tk < nc; ++tk) {
#pragma unroll
        for (int rr0 = 0; rr0 < 4; ++rr0) {
            const int k0 = 16 * tk + 4 * rr0;
            if (k0 >= nc) break;
            const int bw = nc - k0 < 4 ? nc - k0 : 4;
            // the 4 x 4 diagonal block to every lane: element (a, b) sits in lane (l15 = 4 rr0 + a, l4 = b) of the diagonal
            // tile.  One wave per front: straight from the registers (v_readlane); several waves: through LDS.
            double a00, a10, a11, a20, a21, a22, a30, a31, a32, a33;
            if (NW == 1) {
                double dv = 0.0;
#pragma unroll
                for (int q = 0; q < MAXT; ++q) if (ti_[q] == tk && tj_[q] == tk) dv = acc[q][rr0];
                const int b0 = 4 * rr0;
                a00 = mf_readlane(dv, b0);
                a10 = mf_readlane(dv, b0 + 1); a11 = mf_readlane(dv, b0 + 1 + 16);
                a20 = mf_readlane(dv, b0 + 2); a21 = mf_readlane(dv, b0 + 2 + 16); a22 = mf_readlane(dv, b0 + 2 + 32);
                a30 = mf_readlane(dv, b0 + 3); a31 = mf_readlane(dv, b0 + 3 + 16); a32 = mf_readlane(dv, b0 + 3 + 32);
                a33 = mf_readlane(dv, b0 + 3 + 48);
            } else {
#pragma unroll
                for (int q = 0; q < MAXT; ++q)
                    if (ti_[q] == tk && tj_[q] == tk) {
                        const int a = l15 - 4 * rr0;
                        if (a >= 0 && a < 4) blk[a * 4 + l4] = acc[q][rr0];
                    }
                __syncthreads();
                a00 = blk[0]; a10 = blk[4]; a11 = blk[5]; a20 = blk[8]; a21 = blk[9]; a22 = blk[10]; a30 = blk[12]; a31 = blk[13];
                a32 = blk[14]; a33 = blk[15];
            }
            // the block's LDL^T, same numbers in every lane; columns beyond bw (a partial last block) eliminate nothing
            double i1 = 0.0, i2 = 0.0, i3 = 0.0, l21 = 0.0, l31 = 0.0, l32 = 0.0;
            const double i0 = mf_rcp(a00);
            const double l10 = a10 * i0, l20 = a20 * i0, l30 = a30 * i0;
            a11 -= l10 * a10; a21 -= l10 * a20; a22 -= l20 * a20; a31 -= l10 * a30; a32 -= l20 * a30; a33 -= l30 * a30;
            if (bw > 1) { i1 = mf_rcp(a11); l21 = a21 * i1; l31 = a31 * i1; a22 -= l21 * a21; a32 -= l21 * a31; a33 -= l31 * a31; }
            if (bw > 2) { i2 = mf_rcp(a22); l32 = a32 * i2; a33 -= l32 * a32; }
            if (bw > 3) i3 = mf_rcp(a33);
            const double lc0 = l4 == 1 ? l10 : (l4 == 2 ? l20 : (l4 == 3 ? l30 : 0.0));
            const double lc1 = l4 == 2 ? l21 : (l4 == 3 ? l31 : 0.0);
            const double lc2 = l4 == 3 ? l32 : 0.0;
            const double ic = l4 == 0 ? i0 : (l4 == 1 ? i1 : (l4 == 2 ? i2 : i3));
#pragma unroll
            for (int q = 0; q < MAXT; ++q)
                if (tj_[q] == tk && ti_[q] >= 0) {
                    double x = acc[q][rr0];
                    x -= __shfl(x, l15) * lc0;       // (the lane swaps of mf_rows3 were measured here too: row by row, each broadcast
                    x -= __shfl(x, l15 + 16) * lc1;  //  behind the update before it, they cost this kernel a quarter of its speed)
                    x -= __shfl(x, l15 + 32) * lc2;
                    acc[q][rr0] = x;
                    const int row = 16 * ti_[q] + l15;
                    Xp[l4 * R + row] = x;
                    Lp[l4 * R + row] = row >= k0 + 4 ? x * ic : 0.0;
                }
            if (tid < bw) dl[k0 + tid] = tid == 0 ? i0 : (tid == 1 ? i1 : (tid == 2 ? i2 : i3));
            __syncthreads();
#pragma unroll
            for (int q = 0; q < MAXT; ++q)
                if (ti_[q] >= 0 && tj_[q] >= tk) {
                    const double b = Xp[l4 * R + 16 * ti_[q] + l15];
                    const double a = -Lp[l4 * R + 16 * tj_[q] + l15];
                    acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
                }
        }
    }
    __syncthreads();
    MF_TR(4)
    // 4. results
    double *dinv = mf_dinv(d, inst, cand) + f0, *vv = mf_vv(d, inst, cand) + f0;
#pragma unroll
    for (int q = 0; q < MAXT; ++q) {
        if (ti_[q] < 0) continue;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int col = 16 * tj_[q] + l4 + 4 * rr, row = 16 * ti_[q] + l15;
            const double v = acc[q][rr];
            if (col < nc) {
                if (row > col && row < fs) G[(long)col * ld + row] = v * dl[col];
                else if (row == fs && with_rhs) vv[col] = v * dl[col];
            } else if (col < fs && row >= col && row <= fs) G[(long)col * ld + row] = v;
        }
    }
    for (int k = tid; k < nc; k += NT) dinv[k] = dl[k];
    MF_TR(5)
}

// ---------------------------------------------------------------------------------------------------------------
// The static front kernel (round 3): k_mf_front<T, NW, LDSIMG>, T = 16-row tiles of the front (compile time).
//
// What the trace of k_mf_factor2 showed (scripts/gpu_mf_trace.py, scripts/probes/latency_probe.hip): ~2 us per four-column
// step on a front of four or eight waves, 1 us on a single wave, where the dependent chain of a step -- 4 x 4 block to
// every lane, its LDL^T (43 cycles per reciprocal), three wave shuffles (85 cycles each), one MFMA (81 cycles) -- accounts
// for a third of that.  The rest is code generation: the tiles of a wave are found by runtime predicates over a register
// array (a ladder of exec-mask branches per phase, scalar registers spilled into vector lanes, the accumulators copied
// between the two halves of the register file every step).  Here NOTHING about the tiles is decided at run time:
//   * the front has T tile rows, tile row r belongs to wave r mod NW, every wave runs its own specialisation of the code
//     (template parameter W) in which the loop over the sixteen-column blocks is unrolled: which tile is the diagonal
//     tile, which are solved against it and which receive the update is known to the compiler, every register index is
//     a constant, there is no branch but the two on the number of live columns;
//   * a block of sixteen columns costs two workgroup barriers (four-column steps: eight):
//       A. the wave that owns the diagonal tile factorises it alone, in registers: four sub-steps of four columns --
//          the 4 x 4 diagonal block through a 64-entry LDS scratch of the wave (10 readlanes of a double cost 216
//          cycles, the LDS round trip 130), its LDL^T, the tile's rows times the inverse of the unit 4 x 4 factor (three
//          INDEPENDENT shuffles: 114 cycles instead of 3 x 85), one rank-4 MFMA on the tile itself; it publishes the
//          scaled tile, the four 4 x 4 records and the pivot reciprocals;
//       B. every tile below solves its sixteen columns the same way (per sub-block the 4 x 4 product, then one MFMA
//          against the published diagonal tile for the later columns) and publishes its rows of L; X = L D stays in
//          the registers of the row's owner;
//       C. every tile to the right receives the rank-16 update as four MFMAs, A operand from LDS, B operand from the
//          registers of the row.
// Same arithmetic as k_mf_factor2 up to the order of a few roundings (the substitution inside a 4 x 4 block is a product
// with the block's inverse); the host reference of mfplan.hip holds both to 1e-11.
struct MfBlk4 { double i0, i1, i2, i3, m10, m20, m21, m30, m31, m32; };

// LDL^T of a 4 x 4 diagonal block known to every lane (uniform arithmetic); bw live columns (a partial last block
// eliminates nothing beyond them).  m = strictly lower part of the inverse of the unit factor.
__device__ __forceinline__ MfBlk4 mf_ldl4(double a00, double a10, double a11, double a20, double a21, double a22, double a30,
                                           double a31, double a32, double a33, int bw)
{
    double i1 = 0.0, i2 = 0.0, i3 = 0.0, l21 = 0.0, l31 = 0.0, l32 = 0.0;
    const double i0 = mf_rcp(a00);
    const double l10 = a10 * i0, l20 = a20 * i0, l30 = a30 * i0;
#if SQPHIP_MF_LDL4_FAST
    // (experiment: the products of a column's entries are formed beside the reciprocal, so that one fma -- not a product and an
    //  fma -- separates a pivot's reciprocal from the next pivot)
    {
        const double p11 = a10 * a10, p21 = a10 * a20, p22 = a20 * a20, p31 = a10 * a30, p32 = a20 * a30, p33 = a30 * a30;
        a11 = fma(-p11, i0, a11); a21 = fma(-p21, i0, a21); a22 = fma(-p22, i0, a22); a31 = fma(-p31, i0, a31); a32 = fma(-p32, i0, a32); a33 = fma(-p33, i0, a33);
    }
    if (bw > 1) {
        i1 = mf_rcp(a11); l21 = a21 * i1; l31 = a31 * i1;
        const double q22 = a21 * a21, q32 = a21 * a31, q33 = a31 * a31;
        a22 = fma(-q22, i1, a22); a32 = fma(-q32, i1, a32); a33 = fma(-q33, i1, a33);
    }
    if (bw > 2) { i2 = mf_rcp(a22); l32 = a32 * i2; a33 = fma(-(a32 * a32), i2, a33); }
#else
    a11 -= l10 * a10; a21 -= l10 * a20; a22 -= l20 * a20; a31 -= l10 * a30; a32 -= l20 * a30; a33 -= l30 * a30;
    if (bw > 1) { i1 = mf_rcp(a11); l21 = a21 * i1; l31 = a31 * i1; a22 -= l21 * a21; a32 -= l21 * a31; a33 -= l31 * a31; }
    if (bw > 2) { i2 = mf_rcp(a22); l32 = a32 * i2; a33 -= l32 * a32; }
#endif
    if (bw > 3) i3 = mf_rcp(a33);
    MfBlk4 B;
    B.i0 = i0; B.i1 = i1; B.i2 = i2; B.i3 = i3;
    B.m10 = -l10; B.m21 = -l21; B.m32 = -l32;
    B.m20 = -(l20 + l21 * B.m10);
    B.m31 = -(l31 + l32 * B.m21);
    B.m30 = -(l30 + l31 * B.m10 + l32 * B.m20);
    return B;
}

// rows of a tile through a 4 x 4 sub-block: x <- x L4^-T, the four entries of a row sitting in the lane groups l4 = 0..3.
// The lane-group selection is arithmetic (s1, s2, s3 = 1.0 in the lane group of that number, else 0.0): written as nested
// ?: on l4 the compiler built a tree of exec-mask branches per use.
struct MfLaneSel { double s0, s1, s2, s3; };
__device__ __forceinline__ MfLaneSel mf_lane_sel(int l4)
{
    return {l4 == 0 ? 1.0 : 0.0, l4 == 1 ? 1.0 : 0.0, l4 == 2 ? 1.0 : 0.0, l4 == 3 ? 1.0 : 0.0};
}
__device__ __forceinline__ double mf_sel_ic(const MfLaneSel &S, const MfBlk4 &B)
{
    return fma(S.s3, B.i3, fma(S.s2, B.i2, fma(S.s1, B.i1, S.s0 * B.i0)));
}
__device__ __forceinline__ double mf_apply4(double x, int l15, const MfLaneSel &S, const MfBlk4 &B)
{
    const MfRows V = mf_rows3(x, l15);
    const double v0 = V.r0, v1 = V.r1, v2 = V.r2;
    const double c0 = fma(S.s3, B.m30, fma(S.s2, B.m20, S.s1 * B.m10));
    const double c1 = fma(S.s3, B.m31, S.s2 * B.m21);
    const double c2 = S.s3 * B.m32;
    return fma(c2, v2, fma(c1, v1, fma(c0, v0, x)));
}

__device__ __forceinline__ void wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// LDS of the static front kernel (doubles): [image R x R (LDSIMG), later the rows of L: 16 x R][scaled diagonal tile 256]
// [4 x 10 block records][1 / D: R][one 64-entry scratch per wave]
__host__ __device__ constexpr int mf_front_lds_doubles(int T, int NW, bool ldsimg)
{
    const int R = 16 * T, u = (ldsimg ? R * R : 0) > 16 * R ? R * R : 16 * R;
    return u + 256 + 40 + R + 64 * NW;
}

// What wave W of NW does for one front of T tile rows, in three pieces over the same accumulator registers (the static
// front kernel runs them back to back; the spine kernel puts the hand-over to the parent front between the last two):
// tiles into registers, the unrolled elimination, results.
template <int T, int NW, int W> struct MfTileSet {
    static constexpr int NROWS = W < T ? (T - W + NW - 1) / NW : 0;
    // tiles of my rows: row slot s <-> tile row ti = W + s NW, tiles tj = 0..ti at acc[off(s) + tj]
    static constexpr int NTILES = NROWS > 0 ? NROWS * (2 * W + (NROWS - 1) * NW + 2) / 2 : 1;
};

// (the tiles travel between the pieces as a struct: a reference-to-array parameter of a vector type does not parse)
template <int T, int NW, int W> struct MfAcc { d4 v[MfTileSet<T, NW, W>::NTILES]; };

template <int T, int NW, int W, bool LDSIMG>
__device__ __forceinline__ void mf_front_load(MfAcc<T, NW, W> &A, const double *F, int LD, int fs, int lane)
{
    auto &acc = A.v;
    constexpr int NROWS = MfTileSet<T, NW, W>::NROWS;
    const int l15 = lane & 15, l4 = lane >> 4;
    int o = 0;
#pragma unroll
    for (int s = 0; s < NROWS; ++s) {
        const int ti = W + s * NW;
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int col = 16 * tj + l4 + 4 * rr, row = 16 * ti + l15;
                const bool in = LDSIMG || (row <= fs && col < fs);
                acc[o + tj][rr] = in ? F[col * LD + row] : 0.0;
            }
        o += ti + 1;
    }
}

// (LDSBAR: the barriers of the elimination wait for LDS traffic only -- the spine kernel holds global loads of the NEXT
//  front in flight across the elimination, and a __syncthreads() would wait for them)
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
template <bool LDSBAR> __device__ __forceinline__ void mf_elim_barrier() { if constexpr (LDSBAR) lds_barrier(); else __syncthreads(); }

// L part (columns below nc) of tile column tk of this wave's tiles, out of the registers: final once step tk of the elimination
// has solved the tiles below the diagonal tile (EARLY, below).  The expressions are those of mf_front_store.
template <int T, int NW, int W, bool LDSIMG>
__device__ __forceinline__ void mf_front_store_col(const MfAcc<T, NW, W> &A, int tk, double *G, int ld, int fs, int nc, int with_rhs,
                                                   const double *lds, double *vv, int lane)
{
    const auto &acc = A.v;
    constexpr int R = 16 * T;
    constexpr int U = (LDSIMG ? R * R : 0) > 16 * R ? R * R : 16 * R;
    constexpr int NROWS = MfTileSet<T, NW, W>::NROWS;
    const double *dl = lds + U + 256 + 40;
    const int l15 = lane & 15, l4 = lane >> 4;
    int o = 0;
#pragma unroll
    for (int s = 0; s < NROWS; ++s) {
        const int ti = W + s * NW;
        if (ti >= tk) {
            const bool inside = ti > tk && 16 * ti + 15 < fs;
            if (inside && 16 * tk + 15 < nc) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int col = 16 * tk + l4 + 4 * rr;
                    G[(long)col * ld + 16 * ti + l15] = acc[o + tk][rr] * dl[col];
                }
            } else {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int col = 16 * tk + l4 + 4 * rr, row = 16 * ti + l15;
                    const double v = acc[o + tk][rr];
                    if (col < nc) {
                        if (row > col && row < fs) G[(long)col * ld + row] = v * dl[col];
                        else if (row == fs && with_rhs) vv[col] = v * dl[col];
                    }
                }
            }
        }
        o += ti + 1;
    }
}

// EARLY (round 4): the L part of a tile column goes to the arena as soon as its step of the elimination is through, instead of
// in one burst behind the last step -- the stores (5 000 - 7 000 cycles of issue per 80-column front: cycle stamps) then overlap
// with the remaining steps.  The barriers of the loop must then wait for LDS traffic only (a __syncthreads() waits for the
// stores in flight too).  Same values to the same places.
template <int T, int NW, int W, bool LDSIMG, bool LDSBAR = false, bool EARLY = false>
__device__ __forceinline__ void mf_front_elim(MfAcc<T, NW, W> &A, int nc, double *lds, int lane, int trs, double *G = nullptr, int ld = 0,
                                              int fs = 0, int with_rhs = 0, double *vv = nullptr)
{
    auto &acc = A.v;
    constexpr int R = 16 * T;
    constexpr int U = (LDSIMG ? R * R : 0) > 16 * R ? R * R : 16 * R;
    constexpr int NROWS = MfTileSet<T, NW, W>::NROWS;
    double *Lp = lds, *dtile = lds + U, *rec = dtile + 256, *dl = rec + 40, *dsc = dl + R + 64 * W;
    const int l15 = lane & 15, l4 = lane >> 4;
    const MfLaneSel LS = mf_lane_sel(l4);
#pragma unroll
    for (int tk = 0; tk < T; ++tk) {
        if (16 * tk >= nc) break;
        if (tk == 1) { MF_TRW(9) }
        const int live = nc - 16 * tk;    // columns of this block that are eliminated (>= 1)
        // A. the diagonal tile, by its owner alone
        if (tk % NW == W) {
            constexpr int dummy = 0; (void)dummy;
            const int sd = (tk - W) / NW;                                   // my row slot of tile row tk
            const int od = sd * (2 * W + (sd - 1) * NW + 2) / 2 + tk;       // off(sd) + tk
            d4 dt = acc[od];
            d4 ls = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                if (4 * sub >= live) break;
                const int bw = live - 4 * sub < 4 ? live - 4 * sub : 4, b0 = 4 * sub;
                const double dv = dt[sub];
                // the 4 x 4 block to every lane: element (b0 + r, b0 + c) sits in lane (b0 + r) + 16 c of dv -- ten v_readlane pairs at
                // constant lanes (round 4; through the wave's LDS scratch before: write, wave barrier, ten reads, wave barrier =
                // 408 of the ~1 330 cycles of a block; SQPHIP_MF_SHFL=1 keeps that path)
#if SQPHIP_MF_SHFL
                dsc[lane] = dv;
                wave_sync_lds();
                if (tk == 0 && sub == 0) { MF_TRW(11) }
                const double a00 = dsc[b0], a10 = dsc[b0 + 1], a11 = dsc[b0 + 17], a20 = dsc[b0 + 2], a21 = dsc[b0 + 18],
                             a22 = dsc[b0 + 34], a30 = dsc[b0 + 3], a31 = dsc[b0 + 19], a32 = dsc[b0 + 35], a33 = dsc[b0 + 51];
                wave_sync_lds();
#else
                if (tk == 0 && sub == 0) { MF_TRW(11) }
                const double a00 = mf_readlane(dv, b0), a10 = mf_readlane(dv, b0 + 1), a11 = mf_readlane(dv, b0 + 17),
                             a20 = mf_readlane(dv, b0 + 2), a21 = mf_readlane(dv, b0 + 18), a22 = mf_readlane(dv, b0 + 34),
                             a30 = mf_readlane(dv, b0 + 3), a31 = mf_readlane(dv, b0 + 19), a32 = mf_readlane(dv, b0 + 35),
                             a33 = mf_readlane(dv, b0 + 51);
#endif
#ifdef SQPHIP_MF_TRACE
                if (tk == 0 && sub == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); MF_TRW(12) }
#endif
                const MfBlk4 B = mf_ldl4(a00, a10, a11, a20, a21, a22, a30, a31, a32, a33, bw);
#ifdef SQPHIP_MF_TRACE
                if (tk == 0 && sub == 0) { double tdep = B.m30; asm volatile("v_mov_b64 %0, %0" : "+v"(tdep)); MF_TRW(13) }
#endif
                const double x = mf_apply4(dv, l15, LS, B);
                const double ic = mf_sel_ic(LS, B);
                dt[sub] = x;
                ls[sub] = x * ic;
#ifdef SQPHIP_MF_TRACE
                if (tk == 0 && sub == 0) { double tdep = ls[sub]; asm volatile("v_mov_b64 %0, %0" : "+v"(tdep)); MF_TRW(14) }
#endif
                dt = __builtin_amdgcn_mfma_f64_16x16x4f64(l15 >= b0 + 4 ? -ls[sub] : 0.0, x, dt, 0, 0, 0);
#ifdef SQPHIP_MF_TRACE
                if (tk == 0 && sub == 0) { double tdep = dt[1]; asm volatile("v_mov_b64 %0, %0" : "+v"(tdep)); MF_TRW(15) }
#endif
                if (lane == 0) {
                    double *o = rec + 10 * sub;
                    o[0] = B.i0; o[1] = B.i1; o[2] = B.i2; o[3] = B.i3; o[4] = B.m10; o[5] = B.m20; o[6] = B.m21; o[7] = B.m30;
                    o[8] = B.m31; o[9] = B.m32;
                }
                if (l15 == 0 && l4 < bw) dl[16 * tk + b0 + l4] = ic;          // lane group l4 holds 1 / d of column b0 + l4
            }
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) dtile[64 * sub + lane] = ls[sub];
            acc[od] = dt;
        }
        if (tk == 0) { MF_TRW(3) }
        mf_elim_barrier<LDSBAR || EARLY>();
        if (tk == 0) { MF_TRW(4) }
        // B. my tiles below the diagonal tile: sixteen columns of every row; the rows of L go to LDS, X = L D stays here
        {
            int o = 0;
#pragma unroll
            for (int s = 0; s < NROWS; ++s) {
                const int ti = W + s * NW;
                if (ti > tk) {
                    d4 pt = acc[o + tk];
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub) {
                        if (4 * sub >= live) break;
                        const double *q = rec + 10 * sub;
                        MfBlk4 B;
                        B.i0 = q[0]; B.i1 = q[1]; B.i2 = q[2]; B.i3 = q[3]; B.m10 = q[4]; B.m20 = q[5]; B.m21 = q[6]; B.m30 = q[7];
                        B.m31 = q[8]; B.m32 = q[9];
                        const double x = mf_apply4(pt[sub], l15, LS, B);
                        const double ic = mf_sel_ic(LS, B);
                        pt[sub] = x;
                        Lp[(4 * sub + l4) * R + 16 * ti + l15] = x * ic;
                        if (sub < 3) pt = __builtin_amdgcn_mfma_f64_16x16x4f64(l15 >= 4 * sub + 4 ? -dtile[64 * sub + lane] : 0.0, x, pt, 0, 0, 0);
                    }
                    acc[o + tk] = pt;
                }
                o += ti + 1;
            }
        }
        if (tk == 0) { MF_TRW(5) }
        mf_elim_barrier<LDSBAR || EARLY>();
        if (tk == 0) { MF_TRW(6) }
        // C. rank-16 update of my tiles to the right
        {
            int o = 0;
#pragma unroll
            for (int s = 0; s < NROWS; ++s) {
                const int ti = W + s * NW;
                if (ti > tk) {
                    const d4 xr = acc[o + tk];
#pragma unroll
                    for (int tj = tk + 1; tj <= ti; ++tj)
#pragma unroll
                        for (int sub = 0; sub < 4; ++sub) {
                            if (4 * sub >= live) break;
                            acc[o + tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(-Lp[(4 * sub + l4) * R + 16 * tj + l15], xr[sub], acc[o + tj], 0, 0, 0);
                        }
                }
                o += ti + 1;
            }
        }
        if constexpr (EARLY) mf_front_store_col<T, NW, W, LDSIMG>(A, tk, G, ld, fs, nc, with_rhs, lds, vv, lane);
    }
}

// results: L (scaled), D^-1 L^-1 b, and -- cb_out -- the contribution block with its right-hand-side row (the spine kernel
// hands the block to the parent front in registers instead: mf_front_scatter)
// CB / ldcb: where the contribution block goes -- entry (row, col) of the front to CB[col * ldcb + row]: the front's own
// storage in the arena (CB = G, ldcb = ld), or the spine kernel's LDS staging area in block coordinates
template <int T, int NW, int W, bool LDSIMG, bool EARLY = false>
__device__ __forceinline__ void mf_front_store(const MfAcc<T, NW, W> &A, double *G, int ld, int fs, int nc,
                                               int with_rhs, double *CB, int ldcb, const double *lds, double *dinv, double *vv, int lane)
{
    const auto &acc = A.v;
    constexpr int R = 16 * T;
    constexpr int U = (LDSIMG ? R * R : 0) > 16 * R ? R * R : 16 * R;
    constexpr int NROWS = MfTileSet<T, NW, W>::NROWS;
    const double *dl = lds + U + 256 + 40;
    const int l15 = lane & 15, l4 = lane >> 4;
    int o = 0;
#pragma unroll
    for (int s = 0; s < NROWS; ++s) {
        const int ti = W + s * NW;
#pragma unroll
        for (int tj = 0; tj <= ti; ++tj) {
            // tiles strictly below the diagonal tiles whose rows all lie inside the front and whose columns are all
            // eliminated (L) or all kept (contribution block) need no test per lane: most tiles of a large front
            const bool inside = ti > tj && 16 * ti + 15 < fs;
            if (inside && 16 * tj + 15 < nc) {
                if constexpr (!EARLY) {          // (EARLY: mf_front_store_col has written the L part step by step)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int col = 16 * tj + l4 + 4 * rr;
                        G[(long)col * ld + 16 * ti + l15] = acc[o + tj][rr] * dl[col];
                    }
                }
            } else if (inside && 16 * tj >= nc) {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) CB[(long)(16 * tj + l4 + 4 * rr) * ldcb + 16 * ti + l15] = acc[o + tj][rr];
            } else {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) {
                    const int col = 16 * tj + l4 + 4 * rr, row = 16 * ti + l15;
                    const double v = acc[o + tj][rr];
                    if (col < nc) {
                        if constexpr (!EARLY) {
                            if (row > col && row < fs) G[(long)col * ld + row] = v * dl[col];
                            else if (row == fs && with_rhs) vv[col] = v * dl[col];
                        }
                    } else if (col < fs && row >= col && row <= fs) CB[(long)col * ldcb + row] = v;
                }
            }
        }
        o += ti + 1;
    }
    if (W == 0) for (int k = lane; k < nc; k += 64) dinv[k] = dl[k];
}

template <int T, int NW, int W, bool LDSIMG, bool LDSBAR = false, bool EARLY = false>
__device__ __forceinline__ void mf_front_wave(const double *F, int LD, double *G, int ld, int fs, int nc, int with_rhs,
                                              double *lds, double *dinv, double *vv, int lane, int trs, double *CB, int ldcb)
{
    MF_TRW(0)
    MfAcc<T, NW, W> acc;
    mf_front_load<T, NW, W, LDSIMG>(acc, F, LD, fs, lane);
    MF_TRW(1)
    // the image is dead from here on: its LDS carries the rows of L -- and, EARLY with the image in the arena, its storage the
    // columns of L: this barrier waits for the loads of the image too (the level kernels' __syncthreads())
    mf_elim_barrier<LDSBAR>();
    MF_TRW(2)
    mf_front_elim<T, NW, W, LDSIMG, LDSBAR, EARLY>(acc, nc, lds, lane, trs, G, ld, fs, with_rhs, vv);
    MF_TRW(7)
    mf_elim_barrier<LDSBAR || EARLY>();
    MF_TRW(8)
    mf_front_store<T, NW, W, LDSIMG, EARLY>(acc, G, ld, fs, nc, with_rhs, CB, ldcb, lds, dinv, vv, lane);
    MF_TRW(10)
}

template <int T, int NW, int W, bool LDSIMG>
__device__ __forceinline__ void mf_front_dispatch(int wave, const double *F, int LD, double *G, int ld, int fs, int nc,
                                                  int with_rhs, double *lds, double *dinv, double *vv, int lane, int trs)
{
    if (wave == W) mf_front_wave<T, NW, W, LDSIMG, false, (SQPHIP_MF_EARLY_STORE != 0 && T >= 5)>(F, LD, G, ld, fs, nc, with_rhs, lds, dinv, vv, lane, trs, G, ld);
    else if constexpr (W + 1 < NW) mf_front_dispatch<T, NW, W + 1, LDSIMG>(wave, F, LD, G, ld, fs, nc, with_rhs, lds, dinv, vv, lane, trs);
}

template <int T, int NW, bool LDSIMG>
__global__ __launch_bounds__(64 * NW) void k_mf_front(DV d, int sbegin, int want, int with_rhs)
{
    constexpr int NT = 64 * NW, R = 16 * T;
    int inst, cand;
    if (!mf_candidate(d, want, inst, cand)) return;
    const MfDev &M = d.mf;
    const int s = M.sched[sbegin + blockIdx.x];
    const MfFrontDesc Fd = M.desc[s];
    const int nc = Fd.nc, nr = Fd.nr, fs = nc + nr, ld = fs + 1, f0 = Fd.first;
    const double *arena = mf_arena(d, inst, cand);
    double *G = mf_arena(d, inst, cand) + Fd.off;
    extern __shared__ double mf_lds[];
    double *F = LDSIMG ? mf_lds : G;
    const int LD = LDSIMG ? R : ld;
    const int tid = threadIdx.x;
    MF_TR(0)
    // image: zero, assembled values, right-hand-side row, contribution blocks of the children.  Everything that comes
    // from global memory -- the front's assembled values with their destinations, the receiving entries of the extend-add
    // with the first source of each, the right-hand side -- is requested BEFORE the image is zeroed (the first PV / PE
    // entries of every thread live in registers meanwhile): one exposed memory round trip per front instead of three
    // (a front is a chain of dependent round trips of 1 - 2 us each; the arithmetic in between is short).
    constexpr int PV = 4, PE = 8;
    int vrc[PV]; double vval[PV]; int ew[PE], eb[PE], ee[PE]; double ea[PE];
    {
        const double *vals = mf_vals(d, inst, cand);
#pragma unroll
        for (int k = 0; k < PV; ++k) {
            const int e = Fd.asm_begin + tid + k * NT;
            vrc[k] = -1; vval[k] = 0.0;
            if (e < Fd.asm_end) { vrc[k] = M.dest_rc[e]; vval[k] = vals[e]; }
        }
#pragma unroll
        for (int k = 0; k < PE; ++k) {
            const int t = Fd.ea_begin + tid + k * NT;
            ew[k] = -1; eb[k] = 0; ee[k] = 0; ea[k] = 0.0;
            if (t < Fd.ea_end) { const MfGather g = M.ea_ent[t]; ew[k] = g.where; eb[k] = g.src_begin; ee[k] = g.src_end; ea[k] = arena[g.src0]; }
        }
    }
    double brhs = 0.0;
    if (with_rhs && tid < nc) brhs = d.xv[(long)inst * d.Fpad + f0 + tid];
    if (LDSIMG) for (int e = tid; e < R * R; e += NT) F[e] = 0.0;
    else for (int e = tid; e < ld * fs; e += NT) F[e] = 0.0;
    __syncthreads();
    MF_TR(1)
    {
        const double *vals = mf_vals(d, inst, cand);
#pragma unroll
        for (int k = 0; k < PV; ++k) if (vrc[k] >= 0) F[(vrc[k] >> 16) * LD + (vrc[k] & 0xffff)] = vval[k];
        for (int e = Fd.asm_begin + tid + PV * NT; e < Fd.asm_end; e += NT) {
            const int rc = M.dest_rc[e];
            F[(rc >> 16) * LD + (rc & 0xffff)] = vals[e];
        }
        if (with_rhs) {
            if (tid < nc) F[tid * LD + fs] = brhs;
            const double *b = d.xv + (long)inst * d.Fpad + f0;
            for (int j = tid + NT; j < nc; j += NT) F[j * LD + fs] = b[j];
        }
    }
    __syncthreads();
    MF_TR(2)
#pragma unroll
    for (int k = 0; k < PE; ++k)
        if (ew[k] >= 0) {
            double a = ea[k];
            for (int q = eb[k] + 1; q < ee[k]; ++q) a += arena[M.ea_src[q]];        // (most entries have one source)
            F[(ew[k] >> 16) * LD + (ew[k] & 0xffff)] += a;
        }
    for (int t = Fd.ea_begin + tid + PE * NT; t < Fd.ea_end; t += NT) {
        const MfGather g = M.ea_ent[t];
        double a = arena[g.src0];
        for (int q = g.src_begin + 1; q < g.src_end; ++q) a += arena[M.ea_src[q]];
        const int rc = g.where;
        F[(rc >> 16) * LD + (rc & 0xffff)] += a;
    }
    __syncthreads();
    MF_TR(3)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    mf_front_dispatch<T, NW, 0, LDSIMG>(wave, F, LD, G, ld, fs, nc, with_rhs, mf_lds, mf_dinv(d, inst, cand) + f0,
                                        mf_vv(d, inst, cand) + f0, tid & 63, inst == 0 ? s : -1);
    MF_TR(4)
    MF_TR(5)
}

// ---------------------------------------------------------------------------------------------------------------
// The spine of the factorisation in ONE launch (k_mf_spine; plan: mfplan.hip, MfSpineFront; round 4).
//
// The levels of the assembly tree from spine_level up hold one or two fronts per instance each (IEEE-118: levels 3 .. 12, 17
// fronts, 620 of the 2 069 columns): as level launches they were twelve kernel boundaries per sweep, each front a chain of
// descriptor -> values / gather entries -> gathered sources -> image before its elimination could start, and every
// contribution block a trip through the arena.  Here one workgroup of eight waves per instance walks these fronts in order
// (children first) with the front image in LDS:
//   * the elimination is that of the static front kernel (mf_front_wave: same tiles, same arithmetic, same bits as
//     k_mf_front<T, 8, true>);
//   * the contribution block of a front whose parent is the NEXT front of the spine never leaves the chip: out of the
//     accumulator registers into an LDS staging area, from there into the parent's image behind the terms the parent
//     gathers from the arena -- the order the gather lists of every other kernel use too (mfplan.hip, children_of);
//   * what the next front needs from global memory -- its assembled values with their destinations, the receiving entries
//     of its extend-add from the arena -- is REQUESTED before the current front is eliminated and held in registers across
//     the elimination, whose barriers wait for LDS traffic only; the gathered sources are requested when the elimination
//     ends;
//   * L, 1 / D and D^-1 L^-1 b go to the arena exactly where the solve kernels expect them.
constexpr int MF_SP_NW = 8, MF_SP_NT = 64 * MF_SP_NW, MF_SP_PV = 3, MF_SP_PE = 6;
struct MfSpinePre { int vrc[MF_SP_PV]; double vval[MF_SP_PV]; int ew[MF_SP_PE], eb[MF_SP_PE], ee[MF_SP_PE], es[MF_SP_PE]; double ea[MF_SP_PE]; double brhs; };

template <int T, int W>
__device__ __forceinline__ void mf_spine_dispatch_w(int wave, double *lds, double *G, int ld, int fs, int nc, int with_rhs, double *dinv,
                                                    double *vv, int lane, double *CB, int ldcb)
{
    if (wave == W) mf_front_wave<T, MF_SP_NW, W, true, true>(lds, 16 * T, G, ld, fs, nc, with_rhs, lds, dinv, vv, lane, -1, CB, ldcb);
    else if constexpr (W + 1 < MF_SP_NW) mf_spine_dispatch_w<T, W + 1>(wave, lds, G, ld, fs, nc, with_rhs, dinv, vv, lane, CB, ldcb);
}
template <int T>
__device__ __forceinline__ void mf_spine_dispatch_t(int Tf, int wave, double *lds, double *G, int ld, int fs, int nc, int with_rhs, double *dinv,
                                                    double *vv, int lane, double *CB, int ldcb)
{
    if (Tf == T) mf_spine_dispatch_w<T, 0>(wave, lds, G, ld, fs, nc, with_rhs, dinv, vv, lane, CB, ldcb);
    else if constexpr (T < 8) mf_spine_dispatch_t<T + 1>(Tf, wave, lds, G, ld, fs, nc, with_rhs, dinv, vv, lane, CB, ldcb);
}

// LDS (doubles): [static front kernels' layout for the tallest front, eight waves][staging area of a block that is handed to the
// parent: (nr + 1) x nr, block coordinates][two row maps of 136 ints]
__global__ __launch_bounds__(MF_SP_NT) void k_mf_spine(DV d, int want, int with_rhs)
{
    constexpr int NT = MF_SP_NT, PV = MF_SP_PV, PE = MF_SP_PE;
    int inst, cand;
    if (!mf_candidate(d, want, inst, cand)) return;
    const MfDev &M = d.mf;
    extern __shared__ double mf_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *arena = mf_arena(d, inst, cand);
    const double *vals = mf_vals(d, inst, cand);
    const double *xv = d.xv + (long)inst * d.Fpad;
    double *dinv = mf_dinv(d, inst, cand), *vv = mf_vv(d, inst, cand);
    double *stage = mf_lds + mf_front_lds_doubles(M.sp_T, MF_SP_NW, true);
    int *relmap = reinterpret_cast<int *>(stage + M.sp_stage);                                          // two maps of 136 ints
    const int n = M.sp_n;
    // a front record into scalar registers (every lane reads the same words)
    auto rec = [&](int k) {
        MfSpineFront R;
        const int *g = reinterpret_cast<const int *>(M.sp_fr + k);
        int *w = reinterpret_cast<int *>(&R);
#pragma unroll
        for (int q = 0; q < 12; ++q) w[q] = __builtin_amdgcn_readfirstlane(g[q]);
        R.pad0 = R.pad1 = R.pad2 = R.pad3 = 0;
        return R;
    };
    // request what front R needs from global memory, first round: destinations + values, receiving entries, right-hand side
    auto issue1 = [&](const MfSpineFront &R, MfSpinePre &P) {
#pragma unroll
        for (int k = 0; k < PV; ++k) {
            const int e = R.asm_begin + tid + k * NT;
            P.vrc[k] = -1; P.vval[k] = 0.0;
            if (e < R.asm_end) { P.vrc[k] = M.dest_rc[e]; P.vval[k] = vals[e]; }
        }
#pragma unroll
        for (int k = 0; k < PE; ++k) {
            const int t = R.ea_begin + tid + k * NT;
            P.ew[k] = -1; P.eb[k] = 0; P.ee[k] = 0; P.es[k] = 0; P.ea[k] = 0.0;
            if (t < R.ea_end) { const MfGather g = M.sp_ent[t]; P.ew[k] = g.where; P.eb[k] = g.src_begin; P.ee[k] = g.src_end; P.es[k] = g.src0; }
        }
        P.brhs = (with_rhs && tid < R.nc) ? xv[R.first + tid] : 0.0;
    };
    // ... second round: the first gathered source of every receiving entry (most have exactly one)
    auto issue2 = [&](MfSpinePre &P) {
#pragma unroll
        for (int k = 0; k < PE; ++k) if (P.ew[k] >= 0) P.ea[k] = arena[P.es[k]];
    };
    // image of front R: zero, then the extend-add from the arena (sums in list order: a = src0 + src1 + ...; image = 0 + a)
    auto begin_image = [&](const MfSpineFront &R, const MfSpinePre &P, int slot) {
        const int Rn = 16 * R.T;
        for (int e = tid; e < Rn * Rn; e += NT) mf_lds[e] = 0.0;
        if (R.handoff && tid <= R.nr) relmap[136 * slot + tid] = M.sp_rel[R.rel + tid];
        lds_barrier();
#pragma unroll
        for (int k = 0; k < PE; ++k)
            if (P.ew[k] >= 0) {
                double a = P.ea[k];
                for (int q = P.eb[k] + 1; q < P.ee[k]; ++q) a += arena[M.sp_src[q]];
                mf_lds[(P.ew[k] >> 16) * Rn + (P.ew[k] & 0xffff)] += a;
            }
        for (int t = R.ea_begin + tid + PE * NT; t < R.ea_end; t += NT) {
            const MfGather g = M.sp_ent[t];
            double a = arena[g.src0];
            for (int q = g.src_begin + 1; q < g.src_end; ++q) a += arena[M.sp_src[q]];
            mf_lds[(g.where >> 16) * Rn + (g.where & 0xffff)] += a;
        }
        lds_barrier();
    };
    // ... finished: the structural entries of the Newton matrix and the right-hand side on top of the children's sums
    auto finish_image = [&](const MfSpineFront &R, const MfSpinePre &P) {
        const int Rn = 16 * R.T, fs = R.nc + R.nr;
#pragma unroll
        for (int k = 0; k < PV; ++k)
            if (P.vrc[k] >= 0) { double *q = mf_lds + (P.vrc[k] >> 16) * Rn + (P.vrc[k] & 0xffff); *q = P.vval[k] + *q; }
        for (int e = R.asm_begin + tid + PV * NT; e < R.asm_end; e += NT) {
            const int rc = M.dest_rc[e];
            double *q = mf_lds + (rc >> 16) * Rn + (rc & 0xffff);
            *q = vals[e] + *q;
        }
        if (with_rhs) {
            if (tid < R.nc) { double *q = mf_lds + tid * Rn + fs; *q = P.brhs + *q; }
            for (int j = tid + NT; j < R.nc; j += NT) { double *q = mf_lds + j * Rn + fs; *q = xv[R.first + j] + *q; }
        }
        lds_barrier();
    };
    MfSpineFront R = rec(0);
    MfSpinePre P;
    issue1(R, P);
    issue2(P);
    begin_image(R, P, 0);
    for (int k = 0; k < n; ++k) {
        MF_TRSP(0)
        finish_image(R, P);
        MF_TRSP(1)
        const bool more = k + 1 < n;
        MfSpineFront Rn = R;
        if (more) { Rn = rec(k + 1); issue1(Rn, P); }          // (held in registers across the elimination: its barriers wait for LDS only)
        MF_TRSP(2)
        const int nc = R.nc, nr = R.nr, fs = nc + nr, ld = fs + 1;
        double *G = arena + R.off;
        // the contribution block: to the arena, or -- its parent is the next front -- to the staging area in block coordinates
        double *CB = R.handoff ? stage - (long)nc * (nr + 1) - nc : G;
        // (the lane index is made opaque per front: with it loop-invariant the compiler hoists the per-lane LDS addresses of all
        //  64 specialisations -- eight front heights x eight waves -- out of this loop and keeps hundreds of them alive: 256
        //  registers and 470 bytes of scratch per lane)
        int lane_k = lane;
        asm volatile("" : "+v"(lane_k));
        mf_spine_dispatch_t<1>(R.T, wave, mf_lds, G, ld, fs, nc, with_rhs, dinv + R.first, vv + R.first, lane_k, CB, R.handoff ? nr + 1 : ld);
        MF_TRSP(3)
        if (!more) break;
        issue2(P);
        // a block that went to the arena is read back by a later front of this workgroup (other waves): have it arrive
        if (!R.handoff) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();                         // every wave is done with 1 / D and the rows of L in LDS; the staged block is complete
        MF_TRSP(4)
        begin_image(Rn, P, (k + 1) & 1);
        MF_TRSP(5)
        if (R.handoff) {
            // the staged block on top of the parent's sums from the arena: entry (r, c) -> (rel[r], rel[c]); rel[nr] = the
            // parent's right-hand-side row.  Every entry has its own destination.
            const int *rel = relmap + 136 * (k & 1);
            const int LDn = 16 * Rn.T;
            for (int e = tid; e < nr * (nr + 1); e += NT) {
                const int c = e / (nr + 1), r = e - c * (nr + 1);
                if (r >= c) mf_lds[rel[c] * LDn + rel[r]] += stage[e];
            }
            lds_barrier();
        }
        MF_TRSP(6)
        R = Rn;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Triangular solves.  One WAVE per front; `y` is the wave's LDS vector (height of the largest front).  Between the
// lanes of one wave LDS accesses are ordered by the hardware; wave_sync() only keeps the compiler from moving them.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// forward substitution of one front: d.xv holds the right-hand side in elimination order; on return d.vv holds
// D^-1 L^-1 b for the columns of the front and the update for the ancestors sits in the last row of the front's
// contribution block (where the fused elimination of k_mf_factor2 leaves it too)
__device__ __forceinline__ void mf_front_fwd(const DV &d, int inst, int s, double *y, int lane, int generic)
{
    const MfDev &M = d.mf;
    const MfFrontDesc Fd = M.desc[s];
    const int nc = Fd.nc, nr = Fd.nr, fs = nc + nr, ld = fs + 1, f0 = Fd.first;
    const int cand = d.ist[inst].sel;
    const double *arena = mf_arena(d, inst, cand);
    double *G = mf_arena(d, inst, cand) + Fd.off;
    const double *b = d.xv + (long)inst * d.Fpad + f0;
    for (int i = lane; i < fs; i += 64) y[i] = i < nc ? b[i] : 0.0;
    wave_sync();
    // updates of the children through the gather lists (one lane per receiving entry, sources in fixed order)
    for (int t = Fd.ev_begin + lane; t < Fd.ev_end; t += 64) {
        const MfGather g = M.ev_ent[t];
        double a = arena[g.src0];
        for (int q = g.src_begin + 1; q < g.src_end; ++q) a += arena[M.ev_src[q]];
        y[g.where] += a;
    }
    wave_sync();
    const double *dinv = mf_dinv(d, inst, cand) + f0;
    double *vv = mf_vv(d, inst, cand) + f0;
    if (nc <= 64 && !generic) {
        // lane i owns y_i of the triangular part; y_k travels by a wave shuffle (no barrier, loads pipeline freely)
        double yi = lane < nc ? y[lane] : 0.0;
#pragma unroll 8
        for (int k = 0; k < nc - 1; ++k) {
            const double l = (lane > k && lane < nc) ? G[(long)k * ld + lane] : 0.0;
            const double yk = mf_readlane(yi, k);
            yi -= l * yk;
        }
        if (lane < nc) { y[lane] = yi; vv[lane] = yi * dinv[lane]; }
        wave_sync();
        for (int i = nc + lane; i < fs; i += 64) {
            double acc = y[i];
#pragma unroll 8
            for (int k = 0; k < nc; ++k) acc -= G[(long)k * ld + i] * y[k];
            G[(long)i * ld + fs] = acc;
        }
    } else {
        for (int k = 0; k < nc; ++k) {
            const double yk = y[k];
            for (int i = k + 1 + lane; i < fs; i += 64) y[i] -= G[(long)k * ld + i] * yk;
            wave_sync();
        }
        for (int k = lane; k < nc; k += 64) vv[k] = y[k] * dinv[k];
        for (int i = nc + lane; i < fs; i += 64) G[(long)i * ld + fs] = y[i];
    }
}

// backward substitution of one front: x_cols = L11^-T (vv_cols - L21' x_rows) into d.xv
__device__ __forceinline__ void mf_front_bwd(const DV &d, int inst, int s, double *x, int lane, int generic)
{
    const MfDev &M = d.mf;
    const MfFrontDesc Fd = M.desc[s];
    const int nc = Fd.nc, nr = Fd.nr, fs = nc + nr, ld = fs + 1, f0 = Fd.first;
    const int cand = d.ist[inst].sel;
    const double *G = mf_arena(d, inst, cand) + Fd.off;
    double *xg = d.xv + (long)inst * d.Fpad;
    const double *vv = mf_vv(d, inst, cand) + f0;
    const int *rows = M.rows + Fd.rowptr;
    for (int i = lane; i < fs; i += 64) x[i] = i < nc ? vv[i] : xg[rows[i - nc]];
    wave_sync();
    if (nc <= 64 && !generic) {
        // lane k owns column k: its own dot product with x_rows, then the unit upper triangular solve by shuffles
        double t = lane < nc ? x[lane] : 0.0;
        if (lane < nc) {
            const double *Gk = G + (long)lane * ld;
#pragma unroll 8
            for (int i = nc; i < fs; ++i) t -= Gk[i] * x[i];
        }
#pragma unroll 8
        for (int i = nc - 1; i > 0; --i) {
            const double l = lane < i ? G[(long)lane * ld + i] : 0.0;
            const double xi = mf_readlane(t, i);
            t -= l * xi;
        }
        if (lane < nc) xg[f0 + lane] = t;
    } else {
        for (int k = nc - 1; k >= 0; --k) {
            double a = 0.0;
            for (int i = k + 1 + lane; i < fs; i += 64) a += G[(long)k * ld + i] * x[i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
            if (lane == 0) x[k] -= a;
            wave_sync();
        }
        for (int k = lane; k < nc; k += 64) xg[f0 + k] = x[k];
    }
}

// A front of more than 64 rows, by the four waves of a workgroup.  The triangular part L11 (its nc x nc corner) is
// staged in LDS with all loads in flight at once; the dependent chain -- 16 columns at a time, the 16 x 16 diagonal block
// by a shuffle chain in wave 0, the rest of the triangle by all threads -- then runs at LDS latency.  The rectangular
// part L21 is one fully parallel pass (its loads do not depend on the chain).  Ls == nullptr (corner too large for the
// LDS of this launch): the same steps straight from the arena.
__device__ __forceinline__ void mf_front_fwd_big(const DV &d, int inst, int s, double *y, double *Ls, int tid)
{
    const MfDev &M = d.mf;
    const MfFrontDesc Fd = M.desc[s];
    const int nc = Fd.nc, nr = Fd.nr, fs = nc + nr, ld = fs + 1, f0 = Fd.first;
    const int cand = d.ist[inst].sel;
    const double *arena = mf_arena(d, inst, cand);
    double *G = mf_arena(d, inst, cand) + Fd.off;
    const double *b = d.xv + (long)inst * d.Fpad + f0, *dinv = mf_dinv(d, inst, cand) + f0;
    double *vv = mf_vv(d, inst, cand) + f0;
    const int lane = tid & 63, wave = tid >> 6;
    if (Ls)
        for (int c = wave; c < nc; c += 4)
            for (int r = c + 1 + lane; r < nc; r += 64) Ls[c * nc + r] = G[(long)c * ld + r];
    double dv[8];                                        // 1 / D of the columns wave 0 finishes: block kb / 16, column lane
#pragma unroll
    for (int q = 0; q < 8; ++q) dv[q] = (wave == 0 && lane < 16 && 16 * q + lane < nc) ? dinv[16 * q + lane] : 0.0;
    for (int i = tid; i < fs; i += 256) y[i] = i < nc ? b[i] : 0.0;
    MF_TRS(1)
    __syncthreads();
    MF_TRS(2)
    for (int t = Fd.ev_begin + tid; t < Fd.ev_end; t += 256) {
        const MfGather g = M.ev_ent[t];
        double a = arena[g.src0];
        for (int q = g.src_begin + 1; q < g.src_end; ++q) a += arena[M.ev_src[q]];
        y[g.where] += a;
    }
    __syncthreads();
    MF_TRS(3)
    const double *L = Ls ? Ls : G;
    const int ll = Ls ? nc : ld;
    for (int kb = 0; kb < nc; kb += 16) {
        const int nb = nc - kb < 16 ? nc - kb : 16;
        if (wave == 0) {
            double l[15];
#pragma unroll
            for (int c = 0; c < 15; ++c) l[c] = (c < nb - 1 && lane > c && lane < nb) ? L[(long)(kb + c) * ll + kb + lane] : 0.0;
            double yi = lane < nb ? y[kb + lane] : 0.0;
#pragma unroll
            for (int c = 0; c < 15; ++c) yi -= l[c] * mf_readlane(yi, c);      // v_readlane: ~20 cycles; __shfl (ds_bpermute): 85
            if (lane < nb) y[kb + lane] = yi;      // (D^-1 y goes to global memory behind the loop: a barrier behind a
        }                                          //  global store waits for the store -- __syncthreads() is vmcnt(0) + s_barrier)
        __syncthreads();
        for (int i = kb + nb + tid; i < nc; i += 256) {          // the rest of the triangle
            double acc = 0.0;
#pragma unroll 16
            for (int c = 0; c < nb; ++c) acc += L[(long)(kb + c) * ll + i] * y[kb + c];
            y[i] -= acc;
        }
        __syncthreads();
    }
    // rows below the supernode: the update for the ancestors, y_i - sum_k L_ik y_k.  A front has few such rows (tens) and
    // many columns: one thread per row would walk nc dependent-latency loads while most of the workgroup idles, so eight
    // column groups share a row (32 rows x 8 groups per pass) and their partial sums meet, in fixed order, in the LDS
    // area the L11 image no longer needs.
    MF_TRS(4)
    if (wave == 0) {                                 // D^-1 L^-1 b of the front's columns, one batch of stores
        if (nc <= 128) {
#pragma unroll
            for (int q = 0; q < 8; ++q) if (lane < 16 && 16 * q + lane < nc) vv[16 * q + lane] = y[16 * q + lane] * dv[q];
        } else for (int k = lane; k < nc; k += 64) vv[k] = y[k] * dinv[k];
    }
    if (Ls && nc >= 16) {
        double *part = Ls;
        const int r = tid & 31, kc = tid >> 5;
        for (int rb = nc; rb < fs; rb += 32) {
            const int i = rb + r;
            double a = 0.0;
            if (i < fs) {
#pragma unroll 4
                for (int k = kc; k < nc; k += 8) a += G[(long)k * ld + i] * y[k];
            }
            part[kc * 32 + r] = a;
            __syncthreads();
            if (kc == 0 && i < fs) {
                double acc = y[i];
#pragma unroll
                for (int q = 0; q < 8; ++q) acc -= part[q * 32 + r];
                G[(long)i * ld + fs] = acc;
            }
            __syncthreads();
        }
        MF_TRS(5)
        return;
    }
    for (int i = nc + tid; i < fs; i += 256) {
        double acc = y[i];
#pragma unroll 8
        for (int k = 0; k < nc; ++k) acc -= G[(long)k * ld + i] * y[k];
        G[(long)i * ld + fs] = acc;
    }
}

__device__ __forceinline__ void mf_front_bwd_big(const DV &d, int inst, int s, double *x, double *Ls, int tid)
{
    const MfDev &M = d.mf;
    const MfFrontDesc Fd = M.desc[s];
    const int nc = Fd.nc, nr = Fd.nr, fs = nc + nr, ld = fs + 1, f0 = Fd.first;
    const int cand = d.ist[inst].sel;
    const double *G = mf_arena(d, inst, cand) + Fd.off;
    double *xg = d.xv + (long)inst * d.Fpad;
    const double *vv = mf_vv(d, inst, cand) + f0;
    const int *rows = M.rows + Fd.rowptr;
    const int lane = tid & 63, wave = tid >> 6;
    double *part = x + fs;                       // 16 block sums behind the vector
    if (Ls)
        for (int c = wave; c < nc; c += 4)
            for (int r = c + 1 + lane; r < nc; r += 64) Ls[c * nc + r] = G[(long)c * ld + r];
    for (int i = tid; i < fs; i += 256) x[i] = i < nc ? vv[i] : xg[rows[i - nc]];
    MF_TRS(9)
    __syncthreads();
    MF_TRS(10)
    {   // x_cols -= L21' x_rows: column k by 8 threads, 32 columns per pass (a front has tens of rows below its columns:
        // more columns in flight per pass matter more than longer coalesced runs), every column independent of the others
        const int c = tid >> 3, r = tid & 7;
        for (int kb = 0; kb < nc; kb += 32) {
            double a = 0.0;
            if (kb + c < nc) {
                const double *Gc = G + (long)(kb + c) * ld;
#pragma unroll 4
                for (int i = nc + r; i < fs; i += 8) a += Gc[i] * x[i];
            }
            a += __shfl_xor(a, 4); a += __shfl_xor(a, 2); a += __shfl_xor(a, 1);
            if (r == 0 && kb + c < nc) x[kb + c] -= a;
        }
    }
    __syncthreads();
    MF_TRS(11)
    const double *L = Ls ? Ls : G;
    const int ll = Ls ? nc : ld;
    for (int kb = ((nc - 1) >> 4) << 4; kb >= 0; kb -= 16) {
        const int nb = nc - kb < 16 ? nc - kb : 16;
        {   // column c of the block by 16 threads: its dot product with the finished x inside the triangle
            const int c = tid >> 4, r = tid & 15;
            double a = 0.0;
            if (c < nb) {
                const double *Lc = L + (long)(kb + c) * ll;
                for (int i = kb + nb + r; i < nc; i += 16) a += Lc[i] * x[i];
            }
            a += __shfl_xor(a, 8); a += __shfl_xor(a, 4); a += __shfl_xor(a, 2); a += __shfl_xor(a, 1);
            if (r == 0) part[c] = a;
        }
        __syncthreads();
        if (wave == 0) {
            double l[15];
#pragma unroll
            for (int c = 1; c < 16; ++c) l[c - 1] = (c < nb && lane < c) ? L[(long)(kb + lane) * ll + kb + c] : 0.0;
            double t = lane < nb ? x[kb + lane] - part[lane] : 0.0;
#pragma unroll
            for (int c = 15; c > 0; --c) t -= l[c - 1] * mf_readlane(t, c);
            if (lane < nb) x[kb + lane] = t;
        }
        __syncthreads();
    }
    for (int k = tid; k < nc; k += 256) xg[f0 + k] = x[k];        // (one batch of stores behind the loop, see the forward routine)
    MF_TRS(12)
}

// level-by-level launches: a workgroup of four waves per work item = one front of more than 64 rows (all four waves)
// or up to four smaller fronts (one wave each; nothing but wave-level synchronisation on that path)
__global__ __launch_bounds__(256) void k_mf_fwd(DV d, int ibegin, int want, int generic, int wstride, int vecsz, int lcap)
{
    const int inst = blockIdx.y;
    if (d.phase[inst] != want) return;
    extern __shared__ double mf_lds[];
    const int4 it = reinterpret_cast<const int4 *>(d.mf.sol_items)[ibegin + blockIdx.x];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (it.y == -2) {
        const int nc = d.mf.desc[it.x].nc;
        mf_front_fwd_big(d, inst, it.x, mf_lds, nc * nc <= lcap ? mf_lds + vecsz : nullptr, threadIdx.x);
        return;
    }
    const int s = wave == 0 ? it.x : (wave == 1 ? it.y : (wave == 2 ? it.z : it.w));
    if (s >= 0) mf_front_fwd(d, inst, s, mf_lds + wstride * wave, threadIdx.x & 63, generic);
}

__global__ __launch_bounds__(256) void k_mf_bwd(DV d, int ibegin, int want, int generic, int wstride, int vecsz, int lcap)
{
    const int inst = blockIdx.y;
    if (d.phase[inst] != want) return;
    extern __shared__ double mf_lds[];
    const int4 it = reinterpret_cast<const int4 *>(d.mf.sol_items)[ibegin + blockIdx.x];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (it.y == -2) {
        const int nc = d.mf.desc[it.x].nc;
        mf_front_bwd_big(d, inst, it.x, mf_lds, nc * nc <= lcap ? mf_lds + vecsz : nullptr, threadIdx.x);
        return;
    }
    const int s = wave == 0 ? it.x : (wave == 1 ? it.y : (wave == 2 ? it.z : it.w));
    if (s >= 0) mf_front_bwd(d, inst, s, mf_lds + wstride * wave, threadIdx.x & 63, generic);
}

// The narrow top of the assembly tree in one launch (mfplan.hip, P.top): one workgroup of four waves per instance walks
// the work items of the levels >= top_level in order -- forward pass upwards, then the backward pass downwards -- with
// a workgroup barrier between items (the vectors and update rows live in global memory, visible to the workgroup
// after the barrier).  The items and the per-front routines are those of k_mf_fwd / k_mf_bwd.
__global__ __launch_bounds__(256) void k_mf_solve_top(DV d, int ibegin, int count, int want, int do_fwd, int generic,
                                                      int wstride, int vecsz, int lcap)
{
    const int inst = blockIdx.x;
    if (d.phase[inst] != want) return;
    extern __shared__ double mf_lds[];
    const int4 *items = reinterpret_cast<const int4 *>(d.mf.sol_items) + ibegin;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (do_fwd)
        for (int i = 0; i < count; ++i) {
            const int4 it = items[i];
            if (it.y == -2) {
                const int nc = d.mf.desc[it.x].nc;
                mf_front_fwd_big(d, inst, it.x, mf_lds, nc * nc <= lcap ? mf_lds + vecsz : nullptr, threadIdx.x);
            } else {
                const int s = wave == 0 ? it.x : (wave == 1 ? it.y : (wave == 2 ? it.z : it.w));
                if (s >= 0) mf_front_fwd(d, inst, s, mf_lds + wstride * wave, threadIdx.x & 63, generic);
            }
            __syncthreads();
        }
    for (int i = count - 1; i >= 0; --i) {
        const int4 it = items[i];
        if (it.y == -2) {
            const int nc = d.mf.desc[it.x].nc;
            mf_front_bwd_big(d, inst, it.x, mf_lds, nc * nc <= lcap ? mf_lds + vecsz : nullptr, threadIdx.x);
        } else {
            const int s = wave == 0 ? it.x : (wave == 1 ? it.y : (wave == 2 ? it.z : it.w));
            if (s >= 0) mf_front_bwd(d, inst, s, mf_lds + wstride * wave, threadIdx.x & 63, generic);
        }
        __syncthreads();
    }
}

// whole solve of one instance by ONE workgroup of NWV waves: the waves deal out the fronts of a level, a workgroup
// barrier closes the level (the vectors live in global memory, visible CU-wide after the barrier).  With hundreds of
// instances in flight this fills the chip without a launch per level: 2 x levels launches become one.
template <int NWV>
__global__ __launch_bounds__(64 * NWV) void k_mf_solve_inst(DV d, int want, int do_fwd, int generic)
{
    const int inst = blockIdx.x;
    if (d.phase[inst] != want) return;
    const MfDev &M = d.mf;
    extern __shared__ double mf_lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *y = mf_lds + (long)wave * M.max_front;
    if (do_fwd)
        for (int l = 0; l < M.nlevels; ++l) {
            for (int q = M.level_ptr[l] + wave; q < M.level_ptr[l + 1]; q += NWV) mf_front_fwd(d, inst, M.level_sn[q], y, lane, generic);
            __syncthreads();
        }
    for (int l = M.nlevels - 1; l >= 0; --l) {
        for (int q = M.level_ptr[l] + wave; q < M.level_ptr[l + 1]; q += NWV) mf_front_bwd(d, inst, M.level_sn[q], y, lane, generic);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The top of the assembly tree, streamed (k_mf_solve_top2; plan: mfplan.hip, MfTopFront).  k_mf_solve_top spends ~12 us
// per front and direction on a chain of dependent memory round trips and workgroup barriers (measured with the cycle
// stamps above: 70 % of a forward visit is the blocked L11 chain, two barriers per sixteen columns, the rest waits for
// global loads).  Here ONE wave does all the arithmetic of a front out of LDS and registers -- lane i owns row i (and
// i + 64) of the front: its right-hand side plus the children's updates, then nc steps y_i -= L_ik y_k with y_k read
// across the wave (v_readlane), which finishes the triangular solve AND the update for the ancestors in one loop; the
// backward pass likewise with lane k owning column k -- while the other three waves fetch the NEXT front's factor image,
// 1 / D, right-hand side and gather lists into the second LDS buffer.  Updates between top fronts, the solution of the
// ancestors and D^-1 L^-1 b stay in LDS; global memory sees only the results (stores nobody waits for).  One
// workgroup barrier per front, and it waits for LDS traffic only.
// (lds_barrier(): defined with the front kernels above)

// What a loader wave holds of a front between requesting it and writing it to LDS: its share of the image of L (rows
// below the diagonal of the nc columns: columns lw, lw + NLW, ...; rows lane and lane + 64), and one of the small
// vectors (1 / D and the right-hand side or D^-1 L^-1 b; the LDS indices of the rows; gather pointers; gather sources).
constexpr int MF_TOP_CH = 28;                    // columns per loader wave (three loaders: fronts of up to 84 columns)
struct MfTopStage { double v0[MF_TOP_CH], v1[MF_TOP_CH], sd0, sd1, sb0, sb1; int si0, si1, sg0, sg1, sg2, ss0, ss1; };

// request: every load of the wave in flight at once, nothing waited for
template <int NLW>
__device__ __forceinline__ void mf_top_issue(const MfDev &M, const MfTopFront &F, const double *arena, const double *dinv,
                                             const double *bsrc, int lw, int lane, MfTopStage &S)
{
    const int nc = F.nc, nr = F.nr, fs = F.nc + F.nr, ld = fs + 1;
    const double *G = arena + F.off;
    const int i0 = lane, i1 = lane + 64;
    S.sd0 = S.sd1 = S.sb0 = S.sb1 = 0.0;
    S.si0 = S.si1 = S.sg0 = S.sg1 = S.sg2 = S.ss0 = S.ss1 = 0;
    if (lw == 0) {
        if (i0 < nc) { S.sd0 = dinv[F.first + i0]; if (bsrc) S.sb0 = bsrc[F.first + i0]; }
        if (i1 < nc) { S.sd1 = dinv[F.first + i1]; if (bsrc) S.sb1 = bsrc[F.first + i1]; }
    }
    if (lw == 1 % NLW) {
        if (i0 < nr) S.si0 = M.top_rows[F.rloc + i0];
        if (i1 < nr) S.si1 = M.top_rows[F.rloc + i1];
    }
    if (lw == 2 % NLW) {
        if (i0 <= fs) S.sg0 = M.top_gptr[F.gptr + i0];
        if (i1 <= fs) S.sg1 = M.top_gptr[F.gptr + i1];
        if (lane + 128 <= fs) S.sg2 = M.top_gptr[F.gptr + lane + 128];
    }
    if (lw == NLW - 1) {
        if (i0 < F.nsrc) S.ss0 = M.top_gsrc[F.gsrc0 + i0];
        if (i1 < F.nsrc) S.ss1 = M.top_gsrc[F.gsrc0 + i1];
    }
    // the image: plain loads at clamped rows (a lane beyond the front re-reads its last row), 32-bit offsets from one
    // base, the only test per column a scalar one -- the loader waves are bound by the instructions they issue, not by
    // the memory behind them (with a predicate per lane and load: ~1100 instructions per front and wave)
    const unsigned r0 = i0 < fs ? i0 : fs - 1, r1 = i1 < fs ? i1 : fs - 1;
    if (fs <= 64) {
#pragma unroll
        for (int q = 0; q < MF_TOP_CH; ++q) {
            const int c = lw + q * NLW;
            S.v0[q] = 0.0; S.v1[q] = 0.0;
            if (c < nc) S.v0[q] = G[(unsigned)(c * ld) + r0];
        }
    } else {
#pragma unroll
        for (int q = 0; q < MF_TOP_CH; ++q) {
            const int c = lw + q * NLW;
            S.v0[q] = 0.0; S.v1[q] = 0.0;
            if (c < nc) { S.v0[q] = G[(unsigned)(c * ld) + r0]; S.v1[q] = G[(unsigned)(c * ld) + r1]; }
        }
    }
}
// ... and write what has arrived to the front's LDS buffer (zeros on and above the diagonal: the chains read unmasked)
template <int NLW>
__device__ __forceinline__ void mf_top_commit(const MfDev &M, const MfTopFront &F, double *Bf, bool with_b, int lw, int lane,
                                              const MfTopStage &S)
{
    const int nc = F.nc, nr = F.nr, fs = F.nc + F.nr, ll = F.ll;
    double *dv = Bf + nc * ll, *bv = dv + nc;
    int *gp = reinterpret_cast<int *>(bv + nc), *gs = gp + fs + 1, *rl = gs + F.nsrc;
    const int i0 = lane, i1 = lane + 64;
    const int r0 = i0 < fs ? i0 : fs - 1, r1 = i1 < fs ? i1 : fs - 1;      // (lanes beyond the front write their last row again)
    if (fs <= 64) {
#pragma unroll
        for (int q = 0; q < MF_TOP_CH; ++q) {
            const int c = lw + q * NLW;
            if (c < nc) Bf[c * ll + r0] = r0 > c ? S.v0[q] : 0.0;
        }
    } else {
#pragma unroll
        for (int q = 0; q < MF_TOP_CH; ++q) {
            const int c = lw + q * NLW;
            if (c < nc) { Bf[c * ll + r0] = r0 > c ? S.v0[q] : 0.0; Bf[c * ll + r1] = r1 > c ? S.v1[q] : 0.0; }
        }
    }
    if (lw == 0) {
        if (i0 < nc) { dv[i0] = S.sd0; if (with_b) bv[i0] = S.sb0; }
        if (i1 < nc) { dv[i1] = S.sd1; if (with_b) bv[i1] = S.sb1; }
    }
    if (lw == 1 % NLW) {
        if (i0 < nr) rl[i0] = S.si0;
        if (i1 < nr) rl[i1] = S.si1;
    }
    if (lw == 2 % NLW) {
        if (i0 <= fs) gp[i0] = S.sg0;
        if (i1 <= fs) gp[i1] = S.sg1;
        if (lane + 128 <= fs) gp[lane + 128] = S.sg2;
    }
    if (lw == NLW - 1) {
        if (i0 < F.nsrc) gs[i0] = S.ss0;
        if (i1 < F.nsrc) gs[i1] = S.ss1;
        for (int i = lane + 128; i < F.nsrc; i += 64) gs[i] = M.top_gsrc[F.gsrc0 + i];      // (more than 128 update sources: rare)
    }
}

// A dependent chain, eight steps at a time, software-pipelined by hand: the LDS operands of the next eight steps are
// requested before the current eight run, so a step costs its v_readlane + v_fma_f64 (the dependent v_fma_f64 alone is
// 32 cycles on gfx950, with the v_readlane pair 64: scripts/probes/readlane_chain.hip), not an LDS round trip (left to
// itself the compiler keeps one ds_read per iteration right in front of its use: ~150 cycles per step).
// load(s0, p0, p1): operands of steps s0 .. s0 + 7; step(s0, p0, p1): those steps.  No branches and no selects on the
// operands inside either (with branches the operand arrays went through scratch; a wave-uniform condition in a select
// becomes a branch around the ds_read with a wait right behind it; per-lane selects cost more VALU issue than the chain
// itself): operands are read unmasked -- the image holds zeros on and above the diagonal -- and a step beyond the end
// of the chain gets a zero PIVOT (a scalar select) and changes nothing.
template <class LoadF, class StepF>
__device__ __forceinline__ void mf_pipe8(int nsteps, LoadF load, StepF step)
{
    double a0[8], a1[8], b0[8], b1[8];
    load(0, a0, a1);
    for (int sb = 0; sb < nsteps; sb += 16) {
        load(sb + 8, b0, b1);
        step(sb, a0, a1);
        load(sb + 16, a0, a1);
        if (sb + 8 < nsteps) step(sb + 8, b0, b1);
    }
}
// lane `k` of x if k < lim, else 0 (wave-uniform k: two v_readlane and two scalar selects)
__device__ __forceinline__ double mf_pivot(double x, int k, int lim)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), k), hi = __builtin_amdgcn_readlane(__double2hiint(x), k);
    return __hiloint2double(k < lim ? hi : 0, k < lim ? lo : 0);
}

// The two chains on an LDS image Bf of a front's L (column-major, leading dimension ll, zeros on and above the diagonal),
// by one wave.  Forward: lane i holds y_i (y0) and y_{i + 64} (y1; TWO: the front has more than 64 rows) of the
// front's fs rows; on return rows < nc hold L11^-1 y, rows >= nc the update y_i - L21 (L11^-1 y).
template <bool TWO>
__device__ __forceinline__ void mf_chain_fwd(const double *Bf, int ll, int nc, int fs, int lane, double &y0, double &y1)
{
    const int i0 = lane, i1 = lane + 64;
    const int r0 = i0 < fs ? i0 : 0, r1 = i1 < fs ? i1 : 0;
    // columns 0 .. 63: y_k lives in y0; rows 64.. (y1) lie below every one of them
    const int n0 = nc < 64 ? nc : 64;
    const double *R0 = Bf + r0, *R1 = Bf + r1;
    mf_pipe8(n0,
        [&](int s0, double (&p0)[8], double (&p1)[8]) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = s0 + q, kk = k < n0 ? k : n0 - 1;
                p0[q] = R0[kk * ll];
                if constexpr (TWO) p1[q] = R1[kk * ll];
            }
        },
        [&](int s0, double (&p0)[8], double (&p1)[8]) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const double yk = mf_pivot(y0, s0 + q, n0);
                y0 -= p0[q] * yk;
                if constexpr (TWO) y1 -= p1[q] * yk;
            }
        });
    // columns 64 ..: y_k lives in y1
    if constexpr (TWO) if (nc > 64)
        mf_pipe8(nc - 64,
            [&](int s0, double (&p0)[8], double (&p1)[8]) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int k = 64 + s0 + q, kk = k < nc ? k : nc - 1;
                    p1[q] = R1[kk * ll];
                }
            },
            [&](int s0, double (&p0)[8], double (&p1)[8]) {
#pragma unroll
                for (int q = 0; q < 8; ++q) y1 -= p1[q] * mf_pivot(y1, s0 + q, nc - 64);
            });
}
// Backward: lane k holds t_k (t0) and t_{k + 64} (t1; TWO: more than 64 columns) = D^-1 L^-1 b of the front's columns
// and x of its rows below (xr0: row lane, xr1: row lane + 64); on return t = L11^-T (t - L21' x_rows).
template <bool TWO>
__device__ __forceinline__ void mf_chain_bwd(const double *Bf, int ll, int nc, int nr, int lane, double xr0, double xr1,
                                             double &t0, double &t1)
{
    const int i0 = lane, i1 = lane + 64;
    const int c0 = i0 < nc ? i0 : 0, c1 = i1 < nc ? i1 : 0;
    const double *L0 = Bf + c0 * ll, *L1 = Bf + c1 * ll;          // lane k reads along its column k (ll odd: no bank conflicts)
    // x_cols -= L21' x_rows: row r of the rows below, x_r read across the wave; four partial sums per column (nothing
    // in this loop depends on the step before but the accumulation itself: 32 cycles per dependent v_fma_f64)
    {
        double a0[4] = {0.0, 0.0, 0.0, 0.0}, a1[4] = {0.0, 0.0, 0.0, 0.0};
        mf_pipe8(nr,
            [&](int s0, double (&p0)[8], double (&p1)[8]) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = s0 + q, rr = r < nr ? r : nr - 1;
                    p0[q] = L0[nc + rr];
                    if constexpr (TWO) p1[q] = L1[nc + rr];
                }
            },
            [&](int s0, double (&p0)[8], double (&p1)[8]) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const double xr = s0 < 64 ? mf_pivot(xr0, s0 + q, nr) : mf_pivot(xr1, s0 + q - 64, nr - 64);     // (blocks of eight never straddle 64)
                    a0[q & 3] += p0[q] * xr;
                    if constexpr (TWO) a1[q & 3] += p1[q] * xr;
                }
            });
        t0 -= (a0[0] + a0[1]) + (a0[2] + a0[3]);
        if constexpr (TWO) t1 -= (a1[0] + a1[1]) + (a1[2] + a1[3]);
    }
    // rows nc - 1 .. 64 of the triangle: x_i lives in t1; columns < 64 (t0) lie left of every one of them
    if constexpr (TWO) if (nc > 64)
        mf_pipe8(nc - 64,
            [&](int s0, double (&p0)[8], double (&p1)[8]) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int i = nc - 1 - (s0 + q), ii = i >= 64 ? i : 64;
                    p0[q] = L0[ii]; p1[q] = L1[ii];
                }
            },
            [&](int s0, double (&p0)[8], double (&p1)[8]) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    // step s reads lane nc - 65 - s of t1: a valid step iff that lane index is >= 0
                    const int l = nc - 65 - (s0 + q);
                    const int lo = __builtin_amdgcn_readlane(__double2loint(t1), l), hi = __builtin_amdgcn_readlane(__double2hiint(t1), l);
                    const double xi = __hiloint2double(l >= 0 ? hi : 0, l >= 0 ? lo : 0);
                    t0 -= p0[q] * xi;
                    t1 -= p1[q] * xi;
                }
            });
    // rows min(nc, 64) - 1 .. 1: x_i lives in t0
    const int h0 = nc < 64 ? nc : 64;
    mf_pipe8(h0 - 1,
        [&](int s0, double (&p0)[8], double (&p1)[8]) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int i = h0 - 1 - (s0 + q), ii = i >= 1 ? i : 1;
                p0[q] = L0[ii];
            }
        },
        [&](int s0, double (&p0)[8], double (&p1)[8]) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int l = h0 - 1 - (s0 + q);
                const int lo = __builtin_amdgcn_readlane(__double2loint(t0), l), hi = __builtin_amdgcn_readlane(__double2hiint(t0), l);
                t0 -= p0[q] * __hiloint2double(l >= 1 ? hi : 0, l >= 1 ? lo : 0);
            }
        });
}

// forward substitution of one top front by one wave; uvec = [updates from below the top | updates of top fronts]
template <bool TWO>
__device__ __forceinline__ void mf_top_fwd(const MfTopFront &F, const double *Bf, double *uvec, int next, double *ytop,
                                           double *vv, int lane, int trj)
{
    const int nc = F.nc, fs = F.nc + F.nr, ll = F.ll;
    const double *dv = Bf + nc * ll, *bv = dv + nc;
    const int *gp = reinterpret_cast<const int *>(bv + nc), *gs = gp + fs + 1;
    const int i0 = lane, i1 = lane + 64;
    const int r0 = i0 < fs ? i0 : 0, r1 = i1 < fs ? i1 : 0;
    double y0 = i0 < nc ? bv[i0] : 0.0, y1 = (TWO && i1 < nc) ? bv[i1] : 0.0;
    {   // updates of the children, sources in list order (both rows of a lane side by side: one chain of LDS round trips)
        const int q0 = gp[r0], e0 = i0 < fs ? gp[r0 + 1] : q0, q1 = TWO ? gp[r1] : 0, e1 = (TWO && i1 < fs) ? gp[r1 + 1] : q1;
        const int cnt = max(e0 - q0, e1 - q1);
        for (int t = 0; t < cnt; ++t) {
            const int sa = q0 + t < e0 ? gs[q0 + t] : -1, sb = (TWO && q1 + t < e1) ? gs[q1 + t] : -1;
            if (sa >= 0) y0 += uvec[sa];
            if (TWO && sb >= 0) y1 += uvec[sb];
        }
    }
    MF_TR2W(8)
    mf_chain_fwd<TWO>(Bf, ll, nc, fs, lane, y0, y1);
    MF_TR2W(9)
    if (i0 < nc) { const double v = y0 * dv[i0]; ytop[F.xloc + i0] = v; vv[F.first + i0] = v; }
    else if (i0 < fs) uvec[next + F.uoff + i0 - nc] = y0;
    if constexpr (TWO) {
        if (i1 < nc) { const double v = y1 * dv[i1]; ytop[F.xloc + i1] = v; vv[F.first + i1] = v; }
        else if (i1 < fs) uvec[next + F.uoff + i1 - nc] = y1;
    }
}

// backward substitution of one top front by one wave: x_cols = L11^-T (vs - L21' x_rows); vs = D^-1 L^-1 b of its columns.
// TWO: more than 64 columns (a second register per lane: column lane + 64)
template <bool TWO>
__device__ __forceinline__ void mf_top_bwd(const MfTopFront &F, const double *Bf, const double *vs, double *xtop, double *xg,
                                           int lane, int trj)
{
    const int nc = F.nc, nr = F.nr, fs = nc + nr, ll = F.ll;
    const double *bv = Bf + nc * ll + nc;
    const int *rl = reinterpret_cast<const int *>(bv + nc) + fs + 1 + F.nsrc;
    const int i0 = lane, i1 = lane + 64;
    double t0 = i0 < nc ? vs[i0] : 0.0, t1 = (TWO && i1 < nc) ? vs[i1] : 0.0;
    const double xr0 = i0 < nr ? xtop[rl[i0]] : 0.0, xr1 = i1 < nr ? xtop[rl[i1]] : 0.0;
    MF_TR2W(10)
    mf_chain_bwd<TWO>(Bf, ll, nc, nr, lane, xr0, xr1, t0, t1);
    MF_TR2W(12)
    if (i0 < nc) { xtop[F.xloc + i0] = t0; xg[F.first + i0] = t0; }
    if constexpr (TWO) if (i1 < nc) { xtop[F.xloc + i1] = t1; xg[F.first + i1] = t1; }
}

__global__ __launch_bounds__(256) void k_mf_solve_top2(DV d, int want, int do_fwd, int inertia)
{
    const int inst = blockIdx.x;
    if (inertia && d.phase[inst] == PH_FACTOR) {
        // the inertia test of this instance's factorisation (round 4: k_inertia's job, a launch less per sweep): pivot signs
        // counted by the four waves, decision by thread 0, published to the workgroup by the barrier
        IpmState &st = d.ist[inst];
        const bool spec = mf_speculates(d, st);
        double c[4] = {0.0, 0.0, 0.0, 0.0};
        inertia_count(d, d.dinv + (long)inst * d.Fpad, spec ? d.dinv1 + (long)inst * d.Fpad : nullptr, 256, c[0], c[1], c[2], c[3]);
        extern __shared__ double mf_lds[];
        double (*ish)[4] = reinterpret_cast<double (*)[4]>(mf_lds);      // (sixteen doubles of the dynamic LDS, free until the solve starts)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) c[q] += __shfl_xor(c[q], o);
            if ((threadIdx.x & 63) == 0) ish[q][threadIdx.x >> 6] = c[q];
        }
        __syncthreads();
        if (threadIdx.x == 0)
            inertia_decide(d, inst, st, spec, (ish[0][0] + ish[0][1]) + (ish[0][2] + ish[0][3]), (ish[1][0] + ish[1][1]) + (ish[1][2] + ish[1][3]),
                           (ish[2][0] + ish[2][1]) + (ish[2][2] + ish[2][3]), (ish[3][0] + ish[3][1]) + (ish[3][2] + ish[3][3]));
        __syncthreads();
    }
    if (d.phase[inst] != want) return;
    const MfDev &M = d.mf;
    extern __shared__ double mf_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cand = d.ist[inst].sel;
    const double *arena = mf_arena(d, inst, cand);
    const double *dinv = mf_dinv(d, inst, cand);
    double *vv = mf_vv(d, inst, cand);
    double *xg = d.xv + (long)inst * d.Fpad;
    const int n = M.top_n, next = M.top_next;
    double *buf0 = mf_lds, *buf1 = buf0 + M.top_buf0, *uvec = buf1 + M.top_buf1 + 128, *xtop = uvec + next + M.top_utotal,
           *ytop = xtop + M.top_xtotal;
    int *recs = reinterpret_cast<int *>(ytop + M.top_xtotal);       // the front records, staged once: [n][16]
    // steps: forward over fronts 0 .. n-1, then backward n-1 .. 0 (the root does both in its step); or backward only
    const int nsteps = do_fwd ? 2 * n - 1 : n;
#define MF_TOP_FRONT_OF(j) (do_fwd ? ((j) < n ? (j) : 2 * n - 2 - (j)) : n - 1 - (j))
    // record k out of LDS into scalar registers (every lane reads the same words)
    auto rec = [&](int k) {
        MfTopFront R;
        int *w = reinterpret_cast<int *>(&R);
#pragma unroll
        for (int q = 0; q < 13; ++q) w[q] = __builtin_amdgcn_readfirstlane(recs[16 * k + q]);
        return R;
    };
    // prologue, one round trip deep: the first front straight from its global record (all four waves fetch), the records
    // and the updates from below the top alongside
    MfTopStage St;
    MfTopFront F;
    {
        const int k0 = MF_TOP_FRONT_OF(0);
        const int *Tg = reinterpret_cast<const int *>(M.top_fr);
        int *w = reinterpret_cast<int *>(&F);
#pragma unroll
        for (int q = 0; q < 13; ++q) w[q] = __builtin_amdgcn_readfirstlane(Tg[16 * k0 + q]);
        int xs[4] = {0, 0, 0, 0};
        if (do_fwd) for (int t = tid, q = 0; t < next && q < 4; t += 256, ++q) xs[q] = M.top_ext[t];
        mf_top_issue<4>(M, F, arena, dinv, do_fwd ? xg : vv, wave, lane, St);
        for (int t = tid; t < 16 * n; t += 256) recs[t] = Tg[t];
        if (do_fwd) {
#pragma unroll
            for (int q = 0; q < 4; ++q) if (tid + 256 * q < next) uvec[tid + 256 * q] = arena[xs[q]];
            for (int t = tid + 1024; t < next; t += 256) uvec[t] = arena[M.top_ext[t]];
        }
        mf_top_commit<4>(M, F, (k0 & 1) ? buf1 : buf0, true, wave, lane, St);
    }
    lds_barrier();
    // the loader waves run one front ahead of the buffer they fill: during step j they write front(j + 1), requested a
    // whole step ago, into the buffer front(j - 1) has left, and request front(j + 2); the round trip of a request
    // (~3 us here: the factor of an instance is cold) overlaps a whole step of arithmetic
    if (wave > 0 && nsteps > 1) {
        const MfTopFront F1 = rec(MF_TOP_FRONT_OF(1));
        mf_top_issue<3>(M, F1, arena, dinv, do_fwd ? (1 < n ? xg : nullptr) : vv, wave - 1, lane, St);
    }
    for (int j = 0; j < nsteps; ++j) {
        const int k = MF_TOP_FRONT_OF(j);
        const int kn = j + 1 < nsteps ? MF_TOP_FRONT_OF(j + 1) : -1;
        double *Bf = (k & 1) ? buf1 : buf0;
        MF_TR2S(0, 4)
        if (wave == 0) {
#ifdef SQPHIP_MF_TRACE
            const int trj = inst == 0 && j < 64 ? (do_fwd ? 64 : 0) + j : -1;
#else
            const int trj = -1;
#endif
            const bool fwd = do_fwd && j < n;
            // (a front of up to 64 rows needs one register per lane, a taller one two: rows / columns i and i + 64)
            const double *vs = do_fwd ? ytop + F.xloc : Bf + F.nc * F.ll + F.nc;
            if (fwd) {
                if (F.nc + F.nr <= 64) mf_top_fwd<false>(F, Bf, uvec, next, ytop, vv, lane, trj);
                else mf_top_fwd<true>(F, Bf, uvec, next, ytop, vv, lane, trj);
            }
            if (!fwd || j == n - 1) {
                if (F.nc <= 64) mf_top_bwd<false>(F, Bf, vs, xtop, xg, lane, trj);
                else mf_top_bwd<true>(F, Bf, vs, xtop, xg, lane, trj);
            }
            if (kn >= 0) F = rec(kn);
        } else if (kn >= 0) {
            const MfTopFront Fn = rec(kn);
            const bool with_b = do_fwd ? j + 1 < n : true;
            mf_top_commit<3>(M, Fn, (kn & 1) ? buf1 : buf0, with_b, wave - 1, lane, St);
            if (j + 2 < nsteps) {
                const MfTopFront F2 = rec(MF_TOP_FRONT_OF(j + 2));
                mf_top_issue<3>(M, F2, arena, dinv, do_fwd ? (j + 2 < n ? xg : nullptr) : vv, wave - 1, lane, St);
            }
        }
        MF_TR2S(1, 5)
        lds_barrier();
        MF_TR2S(2, 6)
    }
#undef MF_TOP_FRONT_OF
}

// ---------------------------------------------------------------------------------------------------------------
// Level launches with the chains of the streamed kernel (k_mf_fwd2 / k_mf_bwd2).  k_mf_fwd / k_mf_bwd walk a front
// straight out of the arena: descriptor, right-hand side, gather entries, their sources, then the columns of L a few
// at a time inside the dependent loop -- six to eight exposed round trips of ~3 us for a front whose arithmetic takes
// one.  Here a wave requests the whole image of its front, the vectors and the gather entries at once (three round
// trips: descriptor, everything, the gathered sources), stages the image in its LDS buffer and runs mf_chain_fwd /
// mf_chain_bwd on it; a front of more than 64 rows is fetched by the four waves of the workgroup and solved by one.
template <int NLW, int CH, bool TWO>
__device__ __forceinline__ void mf_img_load(const double *G, int ld, int nc, int fs, double *Bf, int ll, int lw, int lane)
{
    const int i0 = lane, i1 = lane + 64;
    const int r0 = i0 < fs ? i0 : fs - 1, r1 = i1 < fs ? i1 : fs - 1;
    for (int cb = lw; cb < nc; cb += CH * NLW) {
        double v0[CH], v1[CH];
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int c = cb + q * NLW;
            v0[q] = 0.0; v1[q] = 0.0;
            if (c < nc) { v0[q] = G[(unsigned)(c * ld) + (unsigned)r0]; if constexpr (TWO) v1[q] = G[(unsigned)(c * ld) + (unsigned)r1]; }
        }
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            const int c = cb + q * NLW;
            if (c < nc) { Bf[c * ll + r0] = r0 > c ? v0[q] : 0.0; if constexpr (TWO) Bf[c * ll + r1] = r1 > c ? v1[q] : 0.0; }
        }
    }
}

// HASBIG = false: a launch without fronts of more than 64 rows (the wide bottom levels): the four-wave path is compiled out
// and the kernel fits more waves per CU
template <bool HASBIG>
__global__ __launch_bounds__(256, HASBIG ? 4 : 6) void k_mf_fwd2(DV d, int ibegin, int want, int wimg)
{
    const int inst = blockIdx.y;
    if (d.phase[inst] != want) return;
    extern __shared__ double mf_lds[];
    const MfDev &M = d.mf;
    const int4 it = reinterpret_cast<const int4 *>(M.sol_items)[ibegin + blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool big = HASBIG && it.y == -2;
    const int s = big ? it.x : (wave == 0 ? it.x : (wave == 1 ? it.y : (wave == 2 ? it.z : it.w)));
    if (s < 0) return;                                        // (no workgroup barrier on the wave-per-front path)
    const MfFrontDesc Fd = M.desc[s];
    const int nc = Fd.nc, nr = Fd.nr, fs = nc + nr, ld = fs + 1, f0 = Fd.first, ll = fs | 1;
    const int cand = d.ist[inst].sel;
    const double *arena = mf_arena(d, inst, cand);
    double *G = mf_arena(d, inst, cand) + Fd.off;
    const double *b = d.xv + (long)inst * d.Fpad + f0, *dinv = mf_dinv(d, inst, cand) + f0;
    double *vv = mf_vv(d, inst, cand) + f0;
    double *y = big ? mf_lds : mf_lds + wave * (64 + wimg), *Bf = y + (big ? 128 : 64);
    const int i0 = lane, i1 = lane + 64;
    if (!big || wave == 0) {
        // the vectors and the gather entries ride with the image loads; only the gathered sources wait for their entries
        const int t0 = Fd.ev_begin + i0, t1 = Fd.ev_begin + i1;
        MfGather g0{0, 0, 0, 0}, g1{0, 0, 0, 0};
        const bool h0 = t0 < Fd.ev_end, h1 = big && t1 < Fd.ev_end;
        if (h0) g0 = M.ev_ent[t0];
        if (h1) g1 = M.ev_ent[t1];
        const double b0 = i0 < nc ? b[i0] : 0.0, b1 = (big && i1 < nc) ? b[i1] : 0.0;
        const double d0 = i0 < nc ? dinv[i0] : 0.0, d1 = (big && i1 < nc) ? dinv[i1] : 0.0;
        if (!big) mf_img_load<1, 24, false>(G, ld, nc, fs, Bf, ll, 0, lane);
        else mf_img_load<4, 14, true>(G, ld, nc, fs, Bf, ll, 0, lane);
        if (i0 < fs) y[i0] = b0;
        if (big && i1 < fs) y[i1] = b1;
        wave_sync();
        if (h0) { double a = arena[g0.src0]; for (int q = g0.src_begin + 1; q < g0.src_end; ++q) a += arena[M.ev_src[q]]; y[g0.where] += a; }
        wave_sync();
        if (h1) { double a = arena[g1.src0]; for (int q = g1.src_begin + 1; q < g1.src_end; ++q) a += arena[M.ev_src[q]]; y[g1.where] += a; }
        wave_sync();
        double y0 = i0 < fs ? y[i0] : 0.0, y1 = (big && i1 < fs) ? y[i1] : 0.0;
        if (big) __syncthreads();                             // the image: four waves wrote it
        if (!big) { wave_sync(); mf_chain_fwd<false>(Bf, ll, nc, fs, lane, y0, y1); }
        else mf_chain_fwd<true>(Bf, ll, nc, fs, lane, y0, y1);
        if (i0 < nc) vv[i0] = y0 * d0; else if (i0 < fs) G[(long)i0 * ld + fs] = y0;
        if (big) { if (i1 < nc) vv[i1] = y1 * d1; else if (i1 < fs) G[(long)i1 * ld + fs] = y1; }
    } else {
        mf_img_load<4, 14, true>(G, ld, nc, fs, Bf, ll, wave, lane);
        __syncthreads();
    }
}

template <bool HASBIG>
__global__ __launch_bounds__(256, HASBIG ? 4 : 5) void k_mf_bwd2(DV d, int ibegin, int want, int wimg)
{
    const int inst = blockIdx.y;
    if (d.phase[inst] != want) return;
    extern __shared__ double mf_lds[];
    const MfDev &M = d.mf;
    const int4 it = reinterpret_cast<const int4 *>(M.sol_items)[ibegin + blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool big = HASBIG && it.y == -2;
    const int s = big ? it.x : (wave == 0 ? it.x : (wave == 1 ? it.y : (wave == 2 ? it.z : it.w)));
    if (s < 0) return;
    const MfFrontDesc Fd = M.desc[s];
    const int nc = Fd.nc, nr = Fd.nr, fs = nc + nr, ld = fs + 1, f0 = Fd.first, ll = fs | 1;
    const int cand = d.ist[inst].sel;
    const double *G = mf_arena(d, inst, cand) + Fd.off;
    double *xg = d.xv + (long)inst * d.Fpad;
    const double *vv = mf_vv(d, inst, cand) + f0;
    const int *rows = M.rows + Fd.rowptr;
    double *Bf = (big ? mf_lds : mf_lds + wave * (64 + wimg)) + (big ? 128 : 64);
    const int i0 = lane, i1 = lane + 64;
    if (!big || wave == 0) {
        const int q0 = i0 < nr ? rows[i0] : -1, q1 = (big && i1 < nr) ? rows[i1] : -1;
        double t0 = i0 < nc ? vv[i0] : 0.0, t1 = (big && i1 < nc) ? vv[i1] : 0.0;
        if (!big) mf_img_load<1, 24, false>(G, ld, nc, fs, Bf, ll, 0, lane);
        else mf_img_load<4, 14, true>(G, ld, nc, fs, Bf, ll, 0, lane);
        const double xr0 = q0 >= 0 ? xg[q0] : 0.0, xr1 = q1 >= 0 ? xg[q1] : 0.0;
        if (big) __syncthreads(); else wave_sync();
        if (nc <= 64) mf_chain_bwd<false>(Bf, ll, nc, nr, lane, xr0, xr1, t0, t1);
        else mf_chain_bwd<true>(Bf, ll, nc, nr, lane, xr0, xr1, t0, t1);
        if (i0 < nc) xg[f0 + i0] = t0;
        if (big && i1 < nc) xg[f0 + i1] = t1;
    } else {
        mf_img_load<4, 14, true>(G, ld, nc, fs, Bf, ll, wave, lane);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------------
static int mf_generic_solves()
{
    static const int g = getenv("SQPHIP_MF_GENERIC") ? atoi(getenv("SQPHIP_MF_GENERIC")) : 0;
    return g;
}

// Kernels that ask for more than 64 KB of dynamic LDS need the attribute set on the device they run on: done for every
// context at creation, on the context's device (a process may hold contexts on several devices; the attribute is per
// device -- ADVICE r3), under a lock (contexts may be created from several host threads).
void mf_device_setup(Ctx &C)
{
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    bool ok = true;
    auto big = [&](const void *f) { ok &= hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess; };
    big(reinterpret_cast<const void *>(k_mf_front<6, 4, true>)); big(reinterpret_cast<const void *>(k_mf_front<7, 4, true>));
    big(reinterpret_cast<const void *>(k_mf_front<8, 4, true>)); big(reinterpret_cast<const void *>(k_mf_front<6, 8, true>));
    big(reinterpret_cast<const void *>(k_mf_front<7, 8, true>)); big(reinterpret_cast<const void *>(k_mf_front<8, 8, true>));
    big(reinterpret_cast<const void *>(k_mf_fwd2<true>)); big(reinterpret_cast<const void *>(k_mf_bwd2<true>));
    big(reinterpret_cast<const void *>(k_mf_fwd2<false>)); big(reinterpret_cast<const void *>(k_mf_bwd2<false>));
    big(reinterpret_cast<const void *>(k_mf_solve_top2));
    big(reinterpret_cast<const void *>(k_mf_spine));
    (void)hipGetLastError();
    C.mf_big_lds = ok;
}

void mf_factor(Ctx &C, int want, bool with_rhs, bool values_done)
{
    const DV &d = C.d;
    hipStream_t s = C.stream;
    static const bool v1 = getenv("SQPHIP_MF_V1") != nullptr;      // cross-check: the plain rank-1 kernel for every front
    const int wr = (int)with_rhs;
    const int nb = d.mf.fronts1 && d.spec_mode != 0 ? 2 * d.B : d.B;         // with the second candidate the factor side runs over 2 B "instances"
    // (values_done: the stage kernel that built the right-hand sides has assembled the values too: mf_values_block)
    if (!v1 && !values_done) { C.tm.open(s); hipLaunchKernelGGL(k_mf_values, dim3((d.mf.nnzK + 255) / 256, nb), dim3(256), 0, s, d, want); C.tm.close(KC_VALUES, s); }
    // static front kernels (k_mf_front<T, NW, LDSIMG>) unless SQPHIP_MF_STATIC=0 asks for the generic ones (cross-check)
    const bool stat = !(getenv("SQPHIP_MF_STATIC") && atoi(getenv("SQPHIP_MF_STATIC")) == 0);     // (read per call: tests flip it)
    // the levels below the spine as level launches, the spine (mfplan.hip: spine_level) by one workgroup per instance
    const bool spine = !v1 && d.mf.sp_n > 0 && C.mf_big_lds;
    int li = 0, cls_open = -1;        // (detail timers: one event pair around the launches below the narrow top, one around those of the top)
    for (const MfLaunch &L : C.mfp().fac) {
        if (spine && li++ >= C.mfp().fac_below) break;
        const int cls = L.level >= C.mfp().narrow_level ? KC_FRONTS_TOP : KC_FRONTS_LOW;
        if (cls != cls_open) { if (cls_open >= 0) C.tm.close(cls_open, s); C.tm.open(s); cls_open = cls; }
        const dim3 grid(L.count, nb);
        const int T = L.tiles, R = 16 * T;
        if (v1 || T > 13) { hipLaunchKernelGGL((k_mf_factor<256, true>), grid, dim3(256), 0, s, d, L.begin, want, wr); continue; }
        // generic kernels: dynamic LDS = [image (16 T)^2 when it lives in LDS][panel X and L: 2 x 4 x 16 T][4 x 4 block][1 / D: 16 T]
#define MF_GENERIC(NW, MAXT, IMG) hipLaunchKernelGGL((k_mf_factor2<NW, MAXT, IMG>), grid, dim3(64 * NW), 8 * (size_t)((IMG ? R * R : 0) + 9 * R + 16), s, d, L.begin, want, wr, T)
#define MF_STATIC(TT, NW, IMG) hipLaunchKernelGGL((k_mf_front<TT, NW, IMG>), grid, dim3(64 * NW), 8 * (size_t)mf_front_lds_doubles(TT, NW, IMG), s, d, L.begin, want, wr)
        // ... from four tile rows on: below that the generic kernels are as fast per front and lighter (registers, code
        // size) where thousands of small fronts are in flight.  Measured (QP/s, static from T = 1 / from T = 4 / never):
        // 512 x IEEE-118 5565 / 5507 / 5259, 64 x IEEE-118 1372 / - / 1315, 9241 shape 20.6 / 24.3 / 23.4, IEEE-14 50.9 k /
        // 55.4 k / 53.9 k.  SQPHIP_MF_STATIC_MIN moves the threshold (tests run it at 1 to cover every instantiation).
        // (round 4: on a narrow level -- a handful of fronts, i.e. the levels the spine kernel takes over -- the static
        //  kernels run every front: the level-launch build, SQPHIP_MF_SPINE=0, then gives the bits of the spine kernel)
        const int stat_min = getenv("SQPHIP_MF_STATIC_MIN") ? atoi(getenv("SQPHIP_MF_STATIC_MIN")) : (L.count <= 8 && L.level >= C.mfp().narrow_level ? 1 : 4);
        // four waves instead of two for fronts of four (bit 0) / five (bit 1) tile rows on the levels near the top of the
        // tree (a handful of fronts: latency, not occupancy, is what counts there): 512 x IEEE-118 7 100 -> 7 154 / 7 297 / 7 326
        // QP/s with bit 0 / bit 1 / both, same bits (SQPHIP_MF_NW4=0: two waves everywhere)
        // ... and eight instead of four for six to eight tile rows there: 7 323 -> 7 403 QP/s, same bits (SQPHIP_MF_NW8=0: four)
        const bool nw8 = !(getenv("SQPHIP_MF_NW8") && atoi(getenv("SQPHIP_MF_NW8")) == 0);
        const int nw4 = L.count <= 8 ? (getenv("SQPHIP_MF_NW4") ? atoi(getenv("SQPHIP_MF_NW4")) : 3) : 0;
        const bool big_img = C.mf_big_lds && !(getenv("SQPHIP_MF_BIG_LDSIMG") && atoi(getenv("SQPHIP_MF_BIG_LDSIMG")) == 0);     // (read per call: tests flip it)
        // (round 4: nine to twelve tile rows too -- the fronts of 129 .. 192 rows of the 1354- and 9241-bus shapes --, eight waves,
        //  image in the arena: 1354 buses 549 -> 557 QP/s, 9241 buses 28.5 -> 29.2; SQPHIP_MF_STATIC_MAX = 8 gives them back to the
        //  generic kernel)
        static const int stat_max = getenv("SQPHIP_MF_STATIC_MAX") ? atoi(getenv("SQPHIP_MF_STATIC_MAX")) : 12;
        if (!stat || T > stat_max || T > 12 || T < stat_min) {
            if (T <= 2) MF_GENERIC(1, 3, true);
            else if (T <= 4) MF_GENERIC(2, 5, true);
            else if (T <= 5) MF_GENERIC(4, 4, true);
            else if (T <= 8) MF_GENERIC(4, 9, false);
            else MF_GENERIC(8, 12, false);
            continue;
        }
        switch (T) {
        case 1: MF_STATIC(1, 1, true); break;
        case 2: MF_STATIC(2, 1, true); break;
        case 3: MF_STATIC(3, 1, true); break;
        case 4: if (nw4 & 1) MF_STATIC(4, 4, true); else MF_STATIC(4, 2, true); break;
        case 5: if (nw4 & 2) MF_STATIC(5, 4, true); else MF_STATIC(5, 2, true); break;
        // (six to eight tile rows: the image fits the 160 KB of LDS of gfx950 too -- 74 / 100 / 131 KB -- once more than
        //  64 KB of dynamic LDS has been asked for; only for the handful of fronts of a level near the top of the tree, where
        //  one workgroup per CU is all there is anyway: +0.3 % on 512 x IEEE-118; SQPHIP_MF_BIG_LDSIMG=0: image in the arena)
        case 6: if (big_img && L.count <= 8) { if (nw8) MF_STATIC(6, 8, true); else MF_STATIC(6, 4, true); } else MF_STATIC(6, 4, false); break;
        case 7: if (big_img && L.count <= 8) { if (nw8) MF_STATIC(7, 8, true); else MF_STATIC(7, 4, true); } else MF_STATIC(7, 4, false); break;
        case 8: if (big_img && L.count <= 8) { if (nw8) MF_STATIC(8, 8, true); else MF_STATIC(8, 4, true); } else MF_STATIC(8, 4, false); break;
        case 9: MF_STATIC(9, 8, false); break;
        case 10: MF_STATIC(10, 8, false); break;
        case 11: MF_STATIC(11, 8, false); break;
        default: MF_STATIC(12, 8, false); break;
        }
#undef MF_GENERIC
#undef MF_STATIC
    }
    if (cls_open >= 0) C.tm.close(cls_open, s);
    if (spine) { C.tm.open(s); hipLaunchKernelGGL(k_mf_spine, dim3(1, nb), dim3(MF_SP_NT), (size_t)C.mfp().spine_lds_bytes, s, d, want, wr); C.tm.close(KC_FRONTS_TOP, s); }
    C.mf_factor_launches += spine ? (long)C.mfp().fac_below + 1 : (long)C.mfp().fac.size();
}

bool mf_solve_tests_inertia(const Ctx &C)
{
    static const bool off = getenv("SQPHIP_MF_INERTIA_KERNEL") != nullptr;      // experiment switch: keep k_inertia
    return !off && C.d.sparse && C.d.mf.top_n > 0 && !mf_generic_solves() && !(C.d.B >= (getenv("SQPHIP_MF_INST_SOLVE_MIN") ? atoi(getenv("SQPHIP_MF_INST_SOLVE_MIN")) : (1 << 30)));
}

// x (d.xv) <- K^-1 x through the factors; skip_fwd: d.vv already holds D^-1 L^-1 b (fused into mf_factor)
void mf_solve(Ctx &C, int want, bool skip_fwd, bool inertia)
{
    const DV &d = C.d;
    hipStream_t s = C.stream;
    const int generic = mf_generic_solves();
    // a launch per level, one wave per (front, instance).  The alternative -- ONE workgroup of 16 waves per instance
    // walking all levels behind workgroup barriers (k_mf_solve_inst, batch >= SQPHIP_MF_INST_SOLVE_MIN) -- saves the
    // 2 x levels launches but caps the parallelism at 16 waves per instance: measured 5306 against 5948 QP/s on
    // 512 x IEEE-118, so it is off unless asked for.
    static const int inst_min = getenv("SQPHIP_MF_INST_SOLVE_MIN") ? atoi(getenv("SQPHIP_MF_INST_SOLVE_MIN")) : (1 << 30);
    if (d.B >= inst_min && d.mf.max_front * 8 * 16 <= 64 * 1024) {
        hipLaunchKernelGGL(k_mf_solve_inst<16>, dim3(d.B), dim3(1024), (size_t)d.mf.max_front * 8 * 16, s, d, want, skip_fwd ? 0 : 1, generic);
        return;
    }
    // level launches: the LDS-staged kernels where every front of the level fits them (mfplan.hip: L.wimg >= 0)
    const bool lvl2 = !(getenv("SQPHIP_MF_LEVEL2") && atoi(getenv("SQPHIP_MF_LEVEL2")) == 0);      // (read per call: tests flip it)
    if (!C.mf_big_lds) throw std::string("sqphip: the solve kernels could not be granted 160 KB of dynamic LDS on this device (mf_device_setup)");
    if (!skip_fwd && !C.mfp().fwd.empty()) C.tm.open(s);
    if (!skip_fwd)
        for (const MfLaunch &L : C.mfp().fwd) {
            if (lvl2 && !generic && L.wimg >= 0) {
                if (L.hasbig) hipLaunchKernelGGL(k_mf_fwd2<true>, dim3(L.count, d.B), dim3(256), L.lds2, s, d, L.begin, want, L.wimg);
                else hipLaunchKernelGGL(k_mf_fwd2<false>, dim3(L.count, d.B), dim3(256), L.lds2, s, d, L.begin, want, L.wimg);
            }
            else hipLaunchKernelGGL(k_mf_fwd, dim3(L.count, d.B), dim3(256), L.lds_bytes, s, d, L.begin, want, generic, L.tiles, L.cls, L.lds_bytes / 8 - L.cls);
        }
    if (!skip_fwd && !C.mfp().fwd.empty()) C.tm.close(KC_SOLVE_LEVELS, s);
    const bool has_top = (d.mf.top_n > 0 && !generic) || C.mfp().top.count > 0;
    if (has_top) C.tm.open(s);
    if (d.mf.top_n > 0 && !generic) {
        const size_t lds = (size_t)C.mfp().top2_lds_bytes;
        hipLaunchKernelGGL(k_mf_solve_top2, dim3(d.B), dim3(256), lds, s, d, want, skip_fwd ? 0 : 1, inertia ? 1 : 0);
    } else if (const MfLaunch &T = C.mfp().top; T.count > 0)
        hipLaunchKernelGGL(k_mf_solve_top, dim3(d.B), dim3(256), T.lds_bytes, s, d, T.begin, T.count, want, skip_fwd ? 0 : 1, generic,
                           T.tiles, T.cls, T.lds_bytes / 8 - T.cls);
    if (has_top) C.tm.close(KC_SOLVE_TOP, s);
    if (!C.mfp().bwd.empty()) C.tm.open(s);
    for (const MfLaunch &L : C.mfp().bwd) {
        if (lvl2 && !generic && L.wimg >= 0) {
            if (L.hasbig) hipLaunchKernelGGL(k_mf_bwd2<true>, dim3(L.count, d.B), dim3(256), L.lds2, s, d, L.begin, want, L.wimg);
            else hipLaunchKernelGGL(k_mf_bwd2<false>, dim3(L.count, d.B), dim3(256), L.lds2, s, d, L.begin, want, L.wimg);
        }
        else hipLaunchKernelGGL(k_mf_bwd, dim3(L.count, d.B), dim3(256), L.lds_bytes, s, d, L.begin, want, generic, L.tiles, L.cls, L.lds_bytes / 8 - L.cls);
    }
    if (!C.mfp().bwd.empty()) C.tm.close(KC_SOLVE_LEVELS, s);
}

}  // namespace sqphip
