// ldlt.hip -- batched dense LDL^T (no pivoting) and triangular solves for gfx950 (MI355X).
//
// This is the arithmetic the reference leaves to Ipopt's linear solver (MUMPS / MA57,
// /root/reference/examples/acopf/opf.jl:59-64): the factorisation of the interior-point KKT matrix
// K = [W J'; J -D] of each QP sub-problem (/root/reference/src/algorithms/subproblem_JuMP.jl:178).
//
// Layout: each instance owns a column-major Npad x Npad buffer (ld = Npad, Npad = 64*T); only the
// lower triangle is meaningful, rows/cols >= N are identity padding so every kernel works on whole
// 64x64 tiles.  Two-level right-looking blocked algorithm (ldlt_factor): outer panels of R = 4 sub-panels of 64
// columns; per sub-panel k
//   k_diag_factor     : tile (k,k) -> L_kk, 1/D_k                 4 waves, columns split over waves, LDS broadcast
//   k_panel_trsm_mfma : tiles (i,k), i>k -> L_ik, W_ik = L_ik D   one wave per half tile in MFMA accumulator layout
//   k_colupdate       : tile column k+1 of the same outer panel -= W L'  (left-looking, rank 64 j, MFMA)
// and per outer panel
//   k_trailing        : tiles (i,j), i>=j right of the panel: A_ij -= sum_t W_it L_jt'   rank 256, MFMA, LDS-staged
// k_diag_factor / k_panel_trsm_mfma also carry a right-hand side through the elimination (fused forward solve).
// The bulk update carries 87 % of the N^3/3 flops at N = 2813 and is the kernel the roofline in bench.py is quoted on.
#include "sqphip_internal.hpp"

namespace sqphip {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// k-major 64-wide LDS image: an XOR of 16 on the in-row index for odd k puts the two k-rows read by one 32-lane
// group on disjoint bank halves
__device__ __forceinline__ int swz(int k, int i) { return k * 64 + (i ^ ((k & 1) << 4)); }

// value of lane `lane` (a compile-time constant after unrolling) in every lane: two v_readlane_b32 into scalar
// registers -- no trip through the LDS crossbar as with __shfl / ds_bpermute.  For the substitution chains of the
// triangular solves, where the broadcast IS the critical path.
__device__ __forceinline__ double bcast_lane(double x, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------------------------------------
// tile (k,k): unblocked right-looking LDL^T by one 256-thread workgroup.  Thread (r, W) = (tid & 63,
// tid >> 6) owns row r of the tile restricted to the columns c = W (mod 4): 16 entries in registers.
// Step j: the owner wave of column j publishes that column (d_j L_rj) through LDS, everybody applies
// the rank-1 update to its 16 columns.  The per-wave program is specialised on the compile-time wave
// index W (the kernel dispatches with a wave-uniform switch), so every column test below folds at
// compile time; lane conditions are predicated, not branched.  Arithmetic per entry is the textbook
// order.
template <int W>
__device__ __forceinline__ void diag_factor_wave(double *__restrict__ A, int ld, double *__restrict__ dinv_out,
                                                 double (*colbuf)[64], int r, double *__restrict__ bk)
{
    double a[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = A[(long)(4 * q + W) * ld + r];
    double my_dinv = 0.0;
    // fused forward elimination (wave 0 only): bb = entry r of the right-hand side block; after step j it has
    // received -L_rj y_j, so at the end it is y = L_kk^-1 b
    double bb = (W == 0 && bk) ? bk[r] : 0.0;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        constexpr int dummy = 0; (void)dummy;
        const int jq = j >> 2;
        if ((j & 3) == W) colbuf[j & 1][r] = a[jq];   // d_j L_rj before scaling (row r of column j)
        __syncthreads();
        const double dj = colbuf[j & 1][j];
        // reciprocal by v_rcp_f64 + two Newton steps (off the IEEE-division critical path; <= 1 ulp)
        double dji = __builtin_amdgcn_rcp(dj);
        dji = fma(fma(-dj, dji, 1.0), dji, dji);
        dji = fma(fma(-dj, dji, 1.0), dji, dji);
        const bool below = r > j;
        const double l = colbuf[j & 1][r] * dji;      // L_rj for r > j
        const double lm = below ? l : 0.0;
        if ((j & 3) == W) {
            my_dinv = (r == j) ? dji : my_dinv;
            a[jq] = below ? l : a[jq];
        }
        if (W == 0) bb -= lm * __shfl(bb, j);
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (4 * q + W > j) a[q] -= lm * colbuf[j & 1][4 * q + W];
        // pin this step's updates (see panel_trsm_wave): keeps the scheduler from sinking them across barriers
#pragma unroll
        for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(a[q]));
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int c = 4 * q + W;
        if (c < r) A[(long)c * ld + r] = a[q];
    }
    if ((r & 3) == W) dinv_out[r] = my_dinv;          // set by lane r == j of the owner wave of column j
    if (W == 0 && bk) bk[r] = bb;
}

// bvec / vvec (both or neither): right-hand sides [B][Npad] for the fused forward elimination -- b_k is replaced
// by y_k = L_kk^-1 b_k and v_k = D_k^-1 y_k is stored, exactly what k_fwd_step(k) does for its diagonal tile.
__global__ __launch_bounds__(256) void k_diag_factor(double *__restrict__ K, long strideK, int ld,
                                                    double *__restrict__ dinv, int Npad, int k,
                                                    const int *__restrict__ phase, int want,
                                                    double *__restrict__ bvec, double *__restrict__ vvec)
{
    const int inst = blockIdx.x;
    if (phase && phase[inst] != want) return;
    k += blockIdx.y;                 // grid (B, nk): the nk independent leading tile columns in one launch
    const int r = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *A = K + (long)inst * strideK + (long)(k * 64) * ld + k * 64;   // tile origin
    double *dout = dinv + (long)inst * Npad + k * 64;
    __shared__ double colbuf[2][64];
    double *bk = bvec ? bvec + (long)inst * Npad + k * 64 : nullptr;
    switch (w) {
    case 0: diag_factor_wave<0>(A, ld, dout, colbuf, r, bk); break;
    case 1: diag_factor_wave<1>(A, ld, dout, colbuf, r, bk); break;
    case 2: diag_factor_wave<2>(A, ld, dout, colbuf, r, bk); break;
    default: diag_factor_wave<3>(A, ld, dout, colbuf, r, bk); break;
    }
    if (bvec) {
        __syncthreads();             // dout[] of all four waves and bk[] of wave 0 are visible
        if (w == 0) vvec[(long)inst * Npad + k * 64 + r] = bk[r] * dout[r];
    }
}

// ---------------------------------------------------------------------------------------------
// tiles (i,k), i>k: X L_kk' = A_ik, W = X, L = X D^-1.  One 256-thread workgroup per tile.  Lane l of wave W
// owns rows r2 = l & 31 and r2 + 32 restricted to the columns c = cg (mod 8), cg = 2 W + (l >> 5): 2 x 8
// entries in registers.  Right-looking substitution: at step j the owner lanes publish the final x_j through
// LDS, every lane applies x_c -= x_j L_kk[c][j] to its own columns c > j (same subtraction order per entry as
// the textbook loop).  The kernel is bound by the CU's LDS pipe (the reads of L_kk: one per (j, c) pair and
// wave); with two rows per lane each read feeds two multiply-adds -- the one-row-per-lane layout issued 17
// LDS reads per wave and step, this one 10.  The per-wave program is specialised on the compile-time wave
// index like k_diag_factor; the half-wave index enters only through predication and LDS addresses.
template <int W, int TB>
__device__ __forceinline__ void panel_trsm_wave(const double *__restrict__ Lkk, double *__restrict__ A, int ld,
                                                double *__restrict__ Wout, int Npad, int ntile,
                                                const double *__restrict__ di, double *Ls, double (*xs)[TB][64], int lane,
                                                const double *__restrict__ yk, double (*red)[TB][64])
{
    static_assert(TB == 1 || TB == 2, "one or two row tiles per workgroup");
    const int half = lane >> 5, r2 = lane & 31, cg = 2 * W + half;
    // packed strict lower triangle: column j holds rows c > j at Ls[tri(j) + c - j - 1]
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int c = 8 * q + cg;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = r2 + 32 * h;
            const double v = Lkk[(long)c * ld + r];
            if (r > c) Ls[c * 63 - c * (c - 1) / 2 + r - c - 1] = v;
        }
    }
    // Row tile 0 in xa, row tile 1 (TB == 2) in xb: [h * 8 + q] = row r2 + 32 h, column 8 q + cg.  Two separate
    // 16-entry arrays with explicit code per tile: a [TB][2][8] array indexed by a tile loop stays in scratch.
    // A second tile below the last row tile (ntile == 1) is loaded from whatever follows in the buffer -- the
    // reads stay inside this instance's matrix because k < T-1 -- computed on, and never stored.
    double xa[16], xb[16];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const double *Aq = A + (long)(8 * q + cg) * ld + r2;
        xa[q] = Aq[0]; xa[8 + q] = Aq[32];
        if (TB == 2) { xb[q] = Aq[64]; xb[8 + q] = Aq[96]; }
    }
#pragma unroll
    for (int j = 0; j < 63; ++j) {
        if (((j & 7) >> 1) == W) {                  // the wave that owns column j; its half (j & 1) publishes
            if (half == (j & 1)) {
                xs[j & 1][0][r2] = xa[j >> 3]; xs[j & 1][0][r2 + 32] = xa[8 + (j >> 3)];
                if (TB == 2) { xs[j & 1][1][r2] = xb[j >> 3]; xs[j & 1][1][r2 + 32] = xb[8 + (j >> 3)]; }
            }
        }
        __syncthreads();             // also orders the Ls stores of the prologue before their first use
        const double a0 = xs[j & 1][0][r2], a1 = xs[j & 1][0][r2 + 32];
        double b0 = 0.0, b1 = 0.0;
        if (TB == 2) { b0 = xs[j & 1][1][r2]; b1 = xs[j & 1][1][r2 + 32]; }
        // an opaque zero added to this step's L_kk addresses: without it the scheduler hoists the (read-only
        // after the prologue) L_kk reads of many later steps above the barriers and spills
        int z = 0;
        asm volatile("" : "+v"(z));
        const int tri = j * 63 - j * (j - 1) / 2 - j - 1 + z;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (8 * q + 7 <= j) continue;                       // every column of this group is <= j
            double l;
            if (8 * q > j) {                                    // every column of this group is > j
                l = Ls[tri + 8 * q + cg];
            } else {                                            // the group that contains j: lanes with c > j only
                const bool act = 8 * q + cg > j;
                const double lv = Ls[act ? tri + 8 * q + cg : 0];   // index clamped for the idle lanes
                l = act ? lv : 0.0;
            }
            xa[q] -= a0 * l; xa[8 + q] -= a1 * l;
            if (TB == 2) { xb[q] -= b0 * l; xb[8 + q] -= b1 * l; }
        }
        // pin this step's updates here: the scheduler otherwise sinks multiply-adds of far columns across
        // later barriers and keeps their x_j / L operands alive in scratch
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            asm volatile("" : "+v"(xa[q]));
            if (TB == 2) asm volatile("" : "+v"(xb[q]));
        }
    }
    double pa0 = 0.0, pa1 = 0.0, pb0 = 0.0, pb1 = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int c = 8 * q + cg;
        double *Wq = Wout + (long)c * Npad + r2, *Aq = A + (long)c * ld + r2;
        const double dc = di[c];
        const double yc = yk ? yk[c] : 0.0;
        {
            const double l0 = xa[q] * dc, l1 = xa[8 + q] * dc;
            Wq[0] = xa[q]; Wq[32] = xa[8 + q];
            Aq[0] = l0; Aq[32] = l1;
            pa0 += l0 * yc; pa1 += l1 * yc;
        }
        if (TB == 2) {
            const double l0 = xb[q] * dc, l1 = xb[8 + q] * dc;
            if (ntile > 1) {
                Wq[64] = xb[q]; Wq[96] = xb[8 + q];
                Aq[64] = l0; Aq[96] = l1;
            }
            pb0 += l0 * yc; pb1 += l1 * yc;
        }
    }
    // fused forward elimination: this lane's share of (L_ik y_k)[row], combined over the 8 column groups by
    // the kernel body
    if (yk) {
        red[cg][0][r2] = pa0; red[cg][0][r2 + 32] = pa1;
        if (TB == 2) { red[cg][1][r2] = pb0; red[cg][1][r2 + 32] = pb1; }
    }
}

template <int TB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TB == 1 ? 4 : 2, TB == 1 ? 4 : 2))) void k_panel_trsm(double *__restrict__ K, long strideK, int ld,
                                                   const double *__restrict__ dinv,
                                                   double *__restrict__ Wbuf, int Npad, int k, int T,
                                                   const int *__restrict__ phase, int want,
                                                   double *__restrict__ bvec)
{
    const int inst = blockIdx.y;
    if (phase && phase[inst] != want) return;
    const int i = k + 1 + blockIdx.x * TB;          // first row tile of this workgroup
    const int ntile = T - i < TB ? T - i : TB;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *Kb = K + (long)inst * strideK;
    const double *Lkk = Kb + (long)(k * 64) * ld + k * 64;
    double *A = Kb + (long)(k * 64) * ld + i * 64;
    __shared__ double Ls[2016];      // strict lower triangle of L_kk, packed by columns (15.75 KB)
    __shared__ double xs[2][TB][64];
    __shared__ double red[8][TB][64];   // fused forward elimination: partial products per column group
    double *Wout = Wbuf + (long)inst * Npad * 64 + i * 64;   // Wbuf already points at this sub-panel's slot
    const double *di = dinv + (long)inst * Npad + k * 64;
    // bvec (optional): right-hand sides [B][Npad]; y_k = bvec[k-block] was finalised by k_diag_factor, this
    // workgroup subtracts L_ik y_k from its own block(s) -- what k_fwd_step(k) does for tile row i
    const double *yk = bvec ? bvec + (long)inst * Npad + k * 64 : nullptr;
    switch (w) {
    case 0: panel_trsm_wave<0, TB>(Lkk, A, ld, Wout, Npad, ntile, di, Ls, xs, lane, yk, red); break;
    case 1: panel_trsm_wave<1, TB>(Lkk, A, ld, Wout, Npad, ntile, di, Ls, xs, lane, yk, red); break;
    case 2: panel_trsm_wave<2, TB>(Lkk, A, ld, Wout, Npad, ntile, di, Ls, xs, lane, yk, red); break;
    default: panel_trsm_wave<3, TB>(Lkk, A, ld, Wout, Npad, ntile, di, Ls, xs, lane, yk, red); break;
    }
    if (bvec) {
        __syncthreads();
        if (w == 0) {
            double *bi = bvec + (long)inst * Npad + i * 64;
#pragma unroll
            for (int t = 0; t < TB; ++t)
                if (TB == 1 || t < ntile) {
                    double sum = 0.0;
#pragma unroll
                    for (int g = 0; g < 8; ++g) sum += red[g][t][lane];
                    bi[64 * t + lane] -= sum;
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// MFMA panel solve (the variant ldlt_factor launches).  One WAVE per 32 x 64 half tile of (i,k), two tiles per
// workgroup sharing a k-major LDS image of L_kk.  The wave keeps its half tile in the MFMA accumulator layout
//     X[cb][tr][rr]  <->  row 32 h + 16 tr + l15,  column 16 cb + l4 + 4 rr   (l15 = lane & 15, l4 = lane >> 4)
// (a whole tile per wave needs 234 VGPRs and then finds no room beside the resident bulk-update workgroups:
// 153 us per call instead of 32; half a tile fits in 128)
// and works through the four 16-column blocks cb = 0..3:
//   1. substitution inside the block (15 dependent steps): x_j lives in the lanes with l4 == (j & 3); it is
//      broadcast to the other three lane groups of the same rows by a wave shuffle (no LDS round trip, no
//      barrier -- the whole tile belongs to one wave), every lane then updates its columns > j of the block;
//   2. the finished block updates the blocks to its right on the MFMA pipe: D[c][row] -= sum_k L[c][k] X[row][k]
//      with A = L (from the LDS image) and B = X -- and the accumulator layout of X *is* the B-operand layout
//      (column 4 ks + l4 of block cb is accumulator element rr = ks), so no data moves between the two phases.
// 3/4 of the multiply-adds run on the MFMA pipe, the dependent chain is 60 shuffle steps instead of 63
// barrier + LDS steps.  Arithmetic: the same substitution; the contributions of earlier blocks are summed in MFMA
// order (4 at a time) instead of one by one.  Also performs the fused forward elimination (b_i -= L_ik y_k).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4)))
void k_panel_trsm_mfma(double *__restrict__ K, long strideK, int ld, const double *__restrict__ dinv,
                       double *__restrict__ Wbuf, int Npad, int k, int T, const int *__restrict__ phase, int want,
                       double *__restrict__ bvec, int row0, long strideW, const unsigned char *__restrict__ tmask,
                       int Ts)
{
    const int inst = blockIdx.y;
    if (phase && phase[inst] != want) return;
    // grid (., B, nk): nk independent leading tile columns in one launch, each with its own W slot; their row
    // tiles start at row0 (the tiles between are structurally zero).  bvec must be null then: the fused
    // elimination of different columns would race on the same rows (k_fwd_lead does it afterwards).
    k += blockIdx.z; Wbuf += (long)blockIdx.z * strideW;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int i = row0 + blockIdx.x * 2 + (wave >> 1);    // this wave's row tile ...
    const int rh = (wave & 1) * 32;                       // ... and its half of the rows (2 x 16)
    if (tmask) {                                          // both row tiles of this workgroup structurally zero: nothing to do
        const int i0 = row0 + blockIdx.x * 2;
        const bool a = i0 < T && tmask[(i0 - Ts) * Ts + k], b = i0 + 1 < T && tmask[(i0 + 1 - Ts) * Ts + k];
        if (!a && !b) return;                             // workgroup-uniform, before any barrier
    }
    double *Kb = K + (long)inst * strideK;
    const double *Lkk = Kb + (long)(k * 64) * ld + k * 64;
    __shared__ double Ls[64 * 64];                        // Ls[swz(kcol, c)] = L_kk[c][kcol]  (k-major image)
    {
        const int ii = (tid & 31) * 2, kk0 = tid >> 5;    // 8 columns per pass, 8 passes
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int kk = pass * 8 + kk0;
            *reinterpret_cast<d2 *>(&Ls[swz(kk, ii)]) = *reinterpret_cast<const d2 *>(Lkk + (long)kk * ld + ii);
        }
    }
    __syncthreads();
    if (i >= T) return;                                   // no further barriers below
    // batched leading columns: a (remainder tile, leading tile) block that is structurally zero stays zero -- L is
    // already zero there (k_kkt_assemble) and the update skips it, so nothing to compute or store
    if (tmask && !tmask[(i - Ts) * Ts + k]) return;
    double *A = Kb + (long)(k * 64) * ld + i * 64 + rh;
    double *Wout = Wbuf + (long)inst * Npad * 64 + i * 64 + rh;   // Wbuf already points at this sub-panel's slot
    const double *di = dinv + (long)inst * Npad + k * 64;
    d4 X[4][2];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const double *Ac = A + (long)(16 * cb + l4 + 4 * rr) * ld + l15;
#pragma unroll
            for (int tr = 0; tr < 2; ++tr) X[cb][tr][rr] = Ac[16 * tr];
        }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
        // 1. substitution inside block cb
#pragma unroll
        for (int j = 0; j < 15; ++j) {
            const int J = 16 * cb + j;
            double xj[2];
#pragma unroll
            for (int tr = 0; tr < 2; ++tr) xj[tr] = __shfl(X[cb][tr][j >> 2], l15 | ((j & 3) << 4));
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                if (4 * rr + 3 <= j) continue;                       // columns l4 + 4 rr <= j for every lane
                const bool act = l4 + 4 * rr > j;
                const double lv = Ls[swz(J, 16 * cb + l4 + 4 * rr)];  // L[c][J]
                const double l = act ? lv : 0.0;
#pragma unroll
                for (int tr = 0; tr < 2; ++tr) X[cb][tr][rr] -= xj[tr] * l;
            }
            // pin the step (see panel_trsm_wave)
#pragma unroll
            for (int tr = 0; tr < 2; ++tr) asm volatile("" : "+v"(X[cb][tr]));
        }
        // 2. blocks to the right: X[cb2] -= X[cb] * L[cb2-block][cb-block]'
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int cb2 = cb + 1; cb2 < 4; ++cb2) {
                const double a = -Ls[swz(16 * cb + 4 * ks + l4, 16 * cb2 + l15)];
#pragma unroll
                for (int tr = 0; tr < 2; ++tr)
                    X[cb2][tr] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, X[cb][tr][ks], X[cb2][tr], 0, 0, 0);
            }
        }
    }
    // results: W = X, L = X D^-1; fused forward elimination: part[tr] = (L_ik y_k)[row 16 tr + l15]
    const double *yk = bvec ? bvec + (long)inst * Npad + k * 64 : nullptr;
    double part[2] = {0.0, 0.0};
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int c = 16 * cb + l4 + 4 * rr;
            double *Wc = Wout + (long)c * Npad + l15, *Ac = A + (long)c * ld + l15;
            const double dc = di[c];
            const double yc = yk ? yk[c] : 0.0;
#pragma unroll
            for (int tr = 0; tr < 2; ++tr) {
                const double x = X[cb][tr][rr], lx = x * dc;
                Wc[16 * tr] = x;
                Ac[16 * tr] = lx;
                part[tr] += lx * yc;
            }
        }
    if (bvec) {
        double *bi = bvec + (long)inst * Npad + i * 64 + rh;
#pragma unroll
        for (int tr = 0; tr < 2; ++tr) {
            double p = part[tr];
            p += __shfl_xor(p, 16);
            p += __shfl_xor(p, 32);
            if (l4 == 0) bi[16 * tr + l15] -= p;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Schur update on fp64 MFMA.  A workgroup (4 waves) owns a run of `tpb` consecutive 64x64 tiles (ti,tj),
// tj in [jlo,jhi), ti>=tj, in column-major tile order, and for each of them computes
//   A[ti][tj] -= sum_{t<nsub} W_t[ti] * L[tj][kp+t]'      (nsub 64-wide sub-panels, inner depth 64*nsub)
// The C tile is read and written ONCE per call whatever nsub is: with the two-level blocking of
// ldlt_factor the bulk update does 8*nsub flops per byte of C traffic, which lifts it off the HBM roof.
// MFMA operands are arranged so that the accumulator's lane index runs along i (the contiguous direction
// of the column-major tile): T[jj][ii] = sum_k L[j][k] W[i][k], A-operand = L, B-operand = W.  LDS images
// are k-major (Ls[k][j], Ws[k][i]) exactly as the columns lie in HBM; an XOR of 16 on the in-row index
// for odd k puts the two k-rows read by one 32-lane group on disjoint bank halves.
// Pipeline: the operand tiles of step s+1 (the next sub-panel, or sub-panel 0 of the NEXT tile of the run)
// are fetched into registers while step s is multiplied, so the memory latency of a fetch -- several
// thousand cycles when the whole chip streams -- is paid once per run, not once per tile (a cycle-stamp
// trace of the one-tile-per-workgroup version showed 8 k cycles of prologue + 3.5 k of first-fetch wait
// per 16 k cycles of MFMA work; scripts/probes/trailing_trace.hip).

// cycle stamps of one wave per sampled workgroup (scripts/probes/trailing_trace.hip only)
#ifdef SQPHIP_TRACE_TRAILING
__device__ long long g_trace[2048][16];
__device__ long long g_span[1 << 17][3];    // per workgroup: first stamp, last stamp, (XCC_ID << 32 | HW_ID)
#define SQPHIP_TR(i)                                                                                  \
    if (threadIdx.x == 0) {                                                                           \
        const long long now_ = (long long)__builtin_readcyclecounter();                               \
        if ((blockIdx.x % 61) == 0 && blockIdx.x / 61 < 2048 && (i) < 16) g_trace[blockIdx.x / 61][i] = now_; \
        if (blockIdx.x < (1 << 17)) {                                                                 \
            if ((i) == 0) {                                                                           \
                g_span[blockIdx.x][0] = now_;                                                         \
                g_span[blockIdx.x][2] = (long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32 | \
                                        (unsigned)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); \
            }                                                                                         \
            if ((i) == 15) g_span[blockIdx.x][1] = now_;                                              \
        }                                                                                             \
    }
#else
#define SQPHIP_TR(i)
#endif

// Tile schedule of a Schur-update launch over tile columns [jlo, jhi), rows ti >= tj: index t -> (tj, ti).
// Consecutive workgroups of one instance run on one XCD at about the same time (32 CUs x 4 resident
// workgroups), so the order walks the region in super-tiles -- groups of S tile columns, inside a group first
// the triangular diagonal block, then S-wide x h-high blocks below it, column-major inside a block -- so that
// the ~128 tiles in flight share ~2 x 11 operand panels (2.8 MB at rank 256) through that XCD's 4 MB L2
// instead of streaming one W panel per tile from the fabric.  S = 1 is the plain column-major order.
__device__ __forceinline__ void tile_decode(int t, int jlo, int jhi, int T, int S, int &tj, int &ti)
{
    int c0 = jlo;
    for (;;) {
        const int w = jhi - c0 < S ? jhi - c0 : S;                  // columns of this group
        const int cnt = w * (T - c0) - w * (w - 1) / 2;             // its tiles
        if (t < cnt || c0 + w >= jhi) {
            const int tri = w * (w + 1) / 2;
            if (t < tri) {                                          // diagonal block, column-major
                int c = 0, len = w;
                while (t >= len) { t -= len; ++c; --len; }
                tj = c0 + c; ti = tj + t;
                return;
            }
            t -= tri;
            const int h = w >= 8 ? w : 64 / w;                      // block height: ~64 tiles per block
            const int rb = t / (w * h), rem = t - rb * (w * h);
            const int r0 = c0 + w + rb * h;
            const int hb = T - r0 < h ? T - r0 : h;
            const int cj = rem / hb;
            tj = c0 + cj; ti = r0 + rem - cj * hb;
            return;
        }
        t -= cnt; c0 += w;
    }
}

// KC = k-columns per LDS stage (two stages): 32 -> 64 KB of LDS, two workgroups per CU; 16 -> 32 KB, up to four.
template <int KC, bool LIST = false>
__device__ __forceinline__ void schur_update_run(double *__restrict__ K, long strideK, int ld,
                                                 const double *__restrict__ Wbuf, long strideW, int Npad,
                                                 int T, int kp, int nsub, int wslot, int jlo, int jhi, int S, int ntl,
                                                 int tpb, int nrun, int B, const int *__restrict__ phase, int want,
                                                 const int *__restrict__ pair_ptr = nullptr,
                                                 const int *__restrict__ pair_k = nullptr)
{
    constexpr int SPS = 64 / KC;            // stages per 64-wide sub-panel
    constexpr int NP = KC / 8;              // staging passes: 8 columns per pass
    int inst, run;
    const int bid = blockIdx.x;
    if ((B & 7) == 0) {             // keep one instance's tiles on one XCD (its panels stay in that L2)
        const int xcd = bid & 7, q = bid >> 3;
        inst = (q / nrun) * 8 + xcd;
        run = q % nrun;
    } else {
        inst = bid / nrun;
        run = bid % nrun;
    }
    if (phase && phase[inst] != want) return;
    SQPHIP_TR(0)
    // first tile of the run
    int t = run * tpb;
    const int tend = t + tpb < ntl ? t + tpb : ntl;
    int tj, ti;
    tile_decode(t, jlo, jhi, T, S, tj, ti);

    // two stages of KC k-columns each: [stage][Ls | Ws][KC x 64]
    __shared__ double lds[2][2][KC * 64];
    double *Kb = K + (long)inst * strideK;
    const double *Wb = Wbuf + (long)wslot * strideW + (long)inst * Npad * 64;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int jb = (wave >> 1) * 32, ib = (wave & 1) * 32;
    const int l15 = lane & 15, l4 = lane >> 4;
    // operand staging: thread moves 2 doubles per column, 32 threads cover a column, 8 columns per pass
    const int ii = (tid & 31) * 2, kk0 = tid >> 5;
    d2 lv[NP], wv[NP];
    // LIST: only the sub-panels both tiles couple to (pair list of the tile order; one tile per run)
    const int *kl = nullptr;
    if (LIST) {
        const int pi = (ti - jlo) * (ti - jlo + 1) / 2 + (tj - jlo);
        kl = pair_k + pair_ptr[pi];
        nsub = pair_ptr[pi + 1] - pair_ptr[pi];
        if (nsub == 0) return;               // uniform over the workgroup, before any barrier
    }
    auto fetch = [&](int ftj, int fti, int step) {
        const int sub = LIST ? kl[step / SPS] : step / SPS, off = (step % SPS) * KC;
        const double *Lg = Kb + (long)((kp + sub) * 64 + off) * ld + ftj * 64;              // L[tj][kp+sub]
        const double *Wg = Wb + (long)sub * strideW + (long)off * Npad + fti * 64;          // W_sub[ti]
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {
            const int kk = pass * 8 + kk0;
            lv[pass] = *reinterpret_cast<const d2 *>(Lg + (long)kk * ld + ii);
            wv[pass] = *reinterpret_cast<const d2 *>(Wg + (long)kk * Npad + ii);
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {
            const int kk = pass * 8 + kk0;
            *reinterpret_cast<d2 *>(&lds[buf][0][swz(kk, ii)]) = lv[pass];
            *reinterpret_cast<d2 *>(&lds[buf][1][swz(kk, ii)]) = wv[pass];
        }
    };
    const int nstep = SPS * nsub;           // >= 2
    fetch(tj, ti, 0);
    stash(0);
    fetch(tj, ti, 1);
    __syncthreads();
    SQPHIP_TR(1)
    int buf = 0;
    for (; t < tend; ++t) {
        // the tile after this one
        int ntj = tj, nti = ti;
        if (t + 1 < tend) tile_decode(t + 1, jlo, jhi, T, S, ntj, nti);
        double *Cg = Kb + (long)(tj * 64 + jb) * ld + ti * 64 + ib;
        // acc = C, then acc -= products (negated A operand)
        d4 acc[2][2];
#pragma unroll
        for (int bj = 0; bj < 2; ++bj)
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    acc[bj][bi][rr] = Cg[(long)(bj * 16 + l4 + 4 * rr) * ld + bi * 16 + l15];
        for (int step = 0; step < nstep; ++step) {
            // registers hold (or are receiving) the operands of the step after this one
            const double *Ls = lds[buf][0], *Ws = lds[buf][1];
#pragma unroll 4
            for (int ks = 0; ks < KC / 4; ++ks) {
                const int kk = ks * 4 + l4;
                const double a0 = -Ls[swz(kk, jb + l15)];
                const double a1 = -Ls[swz(kk, jb + 16 + l15)];
                const double b0 = Ws[swz(kk, ib + l15)];
                const double b1 = Ws[swz(kk, ib + 16 + l15)];
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
            }
            if (step + 1 < nstep || t + 1 < tend) {
                stash(buf ^ 1);                 // the other stage: nobody reads it during this step
                // operands two steps ahead
                if (step + 2 < nstep) fetch(tj, ti, step + 2);
                else if (t + 1 < tend) fetch(ntj, nti, step + 2 - nstep);
            }
            __syncthreads();
            buf ^= 1;
            SQPHIP_TR(2 + step + (t + 1 == tend ? 0 : 100))
        }
#pragma unroll
        for (int bj = 0; bj < 2; ++bj)
#pragma unroll
            for (int bi = 0; bi < 2; ++bi)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    Cg[(long)(bj * 16 + l4 + 4 * rr) * ld + bi * 16 + l15] = acc[bj][bi][rr];
        tj = ntj; ti = nti;
    }
    SQPHIP_TR(15)
}

// the bulk updates (head + rest of ldlt_factor): the kernel the roofline is quoted on
template <int KC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_trailing(double *__restrict__ K, long strideK, int ld,
                                                    const double *__restrict__ Wbuf, long strideW, int Npad,
                                                    int T, int kp, int nsub, int wslot, int jlo, int jhi, int S, int ntl,
                                                    int tpb, int nrun, int B, const int *__restrict__ phase, int want)
{
    schur_update_run<KC>(K, strideK, ld, Wbuf, strideW, Npad, T, kp, nsub, wslot, jlo, jhi, S, ntl, tpb, nrun, B, phase, want);
}

// the rank-64 Ts update behind the independent leading tile columns, restricted per tile pair to the leading tiles
// both of them couple to (order.hip: pair lists); timed and counted with k_trailing
template <int KC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_trailing_list(double *__restrict__ K, long strideK, int ld,
                                                    const double *__restrict__ Wbuf, long strideW, int Npad,
                                                    int T, int kp, int nsub, int wslot, int jlo, int jhi, int S, int ntl,
                                                    int tpb, int nrun, int B, const int *__restrict__ phase, int want,
                                                    const int *__restrict__ pair_ptr, const int *__restrict__ pair_k)
{
    schur_update_run<KC, true>(K, strideK, ld, Wbuf, strideW, Npad, T, kp, nsub, wslot, jlo, jhi, S, ntl, tpb, nrun, B, phase, want,
                               pair_ptr, pair_k);
}

// the left-looking updates inside an outer panel (look-ahead stream, not timed)
template <int KC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_colupdate(double *__restrict__ K, long strideK, int ld,
                                                     const double *__restrict__ Wbuf, long strideW, int Npad,
                                                     int T, int kp, int nsub, int wslot, int jlo, int jhi, int S, int ntl,
                                                     int tpb, int nrun, int B, const int *__restrict__ phase, int want)
{
    schur_update_run<KC>(K, strideK, ld, Wbuf, strideW, Npad, T, kp, nsub, wslot, jlo, jhi, S, ntl, tpb, nrun, B, phase, want);
}

// ---------------------------------------------------------------------------------------------
// forward substitution, step k.  Every workgroup of tile row i>=k re-solves the 64x64 unit-lower
// system of tile (k,k) (cheap), then either publishes v_k = D^-1 y_k (i==k) or updates x_i.
__global__ __launch_bounds__(64) void k_fwd_step(const double *__restrict__ K, long strideK, int ld,
                                                const double *__restrict__ dinv,
                                                double *__restrict__ x, double *__restrict__ v,
                                                int Npad, int k, const int *__restrict__ phase, int want, int lead)
{
    const int inst = blockIdx.y;
    if (phase && phase[inst] != want) return;
    // lead: grid (1, B, nk) -- the diagonal solves of the nk independent leading tile columns in one launch; y_k is
    // also stored (into x) for k_fwd_lead; no other workgroup reads x_k in that launch
    if (lead) k += blockIdx.z;
    const int i = k + blockIdx.x;
    const int r = threadIdx.x;
    const double *Kb = K + (long)inst * strideK;
    const double *Lkk = Kb + (long)(k * 64) * ld + k * 64;
    double *xb = x + (long)inst * Npad;
    __shared__ double ys[64];
    double lrow[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) lrow[c] = Lkk[(long)c * ld + r];
    double xr = xb[k * 64 + r];
#pragma unroll
    for (int j = 0; j < 63; ++j) {
        const double yj = bcast_lane(xr, j);
        if (r > j) xr -= lrow[j] * yj;
    }
    if (i == k) {
        v[(long)inst * Npad + k * 64 + r] = xr * dinv[(long)inst * Npad + k * 64 + r];
        if (lead) xb[k * 64 + r] = xr;
        return;
    }
    ys[r] = xr;
    __syncthreads();
    const double *Lik = Kb + (long)(k * 64) * ld + i * 64;
    double acc = 0.0;
#pragma unroll 16
    for (int c = 0; c < 64; ++c) acc += Lik[(long)c * ld + r] * ys[c];
    xb[i * 64 + r] -= acc;
}

// forward elimination of the Ts independent leading tile columns on the rows below them, in one launch:
// x_i -= sum_{k < Ts} L_ik y_k for tile row i >= Ts (y_k = x_k after the diagonal solves).  Wave w takes the tile
// columns k = w (mod 4); lane r owns row r of the tile row.
__global__ __launch_bounds__(256) void k_fwd_lead(const double *__restrict__ K, long strideK, int ld,
                                                 double *__restrict__ x, int Npad, int Ts,
                                                 const int *__restrict__ phase, int want,
                                                 const unsigned char *__restrict__ tmask)
{
    const int inst = blockIdx.y;
    if (phase && phase[inst] != want) return;
    const int i = Ts + blockIdx.x;
    const int r = threadIdx.x & 63, w = threadIdx.x >> 6;
    const double *Kb = K + (long)inst * strideK;
    double *xb = x + (long)inst * Npad;
    __shared__ double part[4][64];
    double acc = 0.0;
    for (int k = w; k < Ts; k += 4) {
        if (tmask && !tmask[(i - Ts) * Ts + k]) continue;      // structurally zero block (wave-uniform test)
        const double *Lik = Kb + (long)(k * 64) * ld + i * 64 + r;
        const double *yk = xb + k * 64;
#pragma unroll 8
        for (int c = 0; c < 64; ++c) acc += Lik[(long)c * ld] * yk[c];
    }
    part[w][r] = acc;
    __syncthreads();
    if (w == 0) xb[i * 64 + r] -= (part[0][r] + part[1][r]) + (part[2][r] + part[3][r]);
}

// backward substitution L' x = v, step k (descending).  Workgroup j<=k re-solves the 64x64 unit-upper
// system of tile (k,k)' (wave 0; the tile is staged transposed-readable in LDS by all four waves);
// j==k publishes x_k, j<k updates v_j -= L_kj' x_k: tile (k,j) is fetched into registers during the
// diagonal solve, transposed through the same LDS buffer, and the row range of the product is split over
// the waves.
__global__ __launch_bounds__(256) void k_bwd_step(const double *__restrict__ K, long strideK, int ld,
                                                 double *__restrict__ x, double *__restrict__ v,
                                                 int Npad, int k, const int *__restrict__ phase, int want, int lead,
                                                 const unsigned char *__restrict__ tmask, int Ts)
{
    const int inst = blockIdx.y;
    if (phase && phase[inst] != want) return;
    // lead: grid (1, B, nk) -- only the diagonal solves x_k = L_kk^-T v_k of the nk independent leading tile columns
    if (lead) k += blockIdx.z;
    const int j = lead ? k : blockIdx.x;
    // a leading tile row that does not couple to tile row k of the remainder: tile (k, j) is structurally zero
    if (tmask && !lead && j < Ts && k >= Ts && !tmask[(k - Ts) * Ts + j]) return;
    const int c = threadIdx.x & 63, w = threadIdx.x >> 6;
    const double *Kb = K + (long)inst * strideK;
    const double *Lkk = Kb + (long)(k * 64) * ld + k * 64;
    double *vb = v + (long)inst * Npad;
    __shared__ double tile[64 * 65];
    __shared__ double xs[64];
    __shared__ double part[4][64];
    // stage tile (k,k) transposed-readable: tile[cc*65 + rr] = L[rr][cc]; wave w moves columns 16w..16w+15
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) tile[(w * 16 + cc) * 65 + c] = Lkk[(long)(w * 16 + cc) * ld + c];
    double lt[16];
    if (j != k) {
        const double *Lkj = Kb + (long)(j * 64 + w * 16) * ld + k * 64 + c;   // tile (k,j): rows k-block, cols j-block
#pragma unroll
        for (int cc = 0; cc < 16; ++cc) lt[cc] = Lkj[(long)cc * ld];
    }
    __syncthreads();
    if (w == 0) {
        double xc = vb[k * 64 + c];
#pragma unroll
        for (int rr = 63; rr > 0; --rr) {
            const double xr = bcast_lane(xc, rr);
            if (c < rr) xc -= tile[c * 65 + rr] * xr;
        }
        if (j == k) x[(long)inst * Npad + k * 64 + c] = xc;
        xs[c] = xc;
    }
    if (j == k) return;
    __syncthreads();
#pragma unroll
    for (int cc = 0; cc < 16; ++cc) tile[(w * 16 + cc) * 65 + c] = lt[cc];
    __syncthreads();
    double acc = 0.0;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) acc += tile[c * 65 + w * 16 + rr] * xs[w * 16 + rr];
    part[w][c] = acc;
    __syncthreads();
    if (w == 0) vb[j * 64 + c] -= (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
}

// ---------------------------------------------------------------------------------------------
// number of tiles (ti >= tj) in tile columns [jlo, jhi)
static int tiles_in_cols(int T, int jlo, int jhi)
{
    int n = 0;
    for (int j = jlo; j < jhi; ++j) n += T - j;
    return n;
}

static void launch_update(const LdltPlan &P, hipStream_t s, double *K, int kp, int nsub, int wslot, int jlo,
                          int jhi, const int *phase, int want, Timers *tm, bool count, bool use_list = false)
{
    if (jhi > P.T) jhi = P.T;
    if (jlo >= jhi) return;
    const int ntl = tiles_in_cols(P.T, jlo, jhi);
    // tiles per workgroup: long runs amortise the pipeline fill, but keep >= ~4 workgroups per CU slot
    int tpb = (int)((long)ntl * P.B / 2048);
    if (tpb > P.tpb_max) tpb = P.tpb_max;
    if (tpb < 1) tpb = 1;
    const int nrun = (ntl + tpb - 1) / tpb;
    const long strideK = (long)P.ld * P.Npad, strideW = (long)P.B * P.Npad * 64;
    std::pair<hipEvent_t, hipEvent_t> ev;
    const bool timed = tm && tm->enabled && count;
    if (timed) { ev = tm->get(); hipEventRecord(ev.first, s); }
    if (use_list && P.pair_ptr) {
        // one tile per workgroup (the pair list belongs to the tile), plain column-major tile order (S = 1): the
        // pair index in the kernel is computed from (ti, tj) directly
        hipLaunchKernelGGL(k_trailing_list<16>, dim3(ntl * P.B), dim3(256), P.trail_pad, s, K, strideK, P.ld, P.Wbuf, strideW,
                           P.Npad, P.T, kp, nsub, wslot, jlo, jhi, P.supertile, ntl, 1, ntl, P.B, phase, want, P.pair_ptr,
                           P.pair_k);
    } else if (count && P.kc == 16)
        hipLaunchKernelGGL(k_trailing<16>, dim3(nrun * P.B), dim3(256), P.trail_pad, s, K, strideK, P.ld, P.Wbuf, strideW, P.Npad,
                           P.T, kp, nsub, wslot, jlo, jhi, P.supertile, ntl, tpb, nrun, P.B, phase, want);
    else if (count)
        hipLaunchKernelGGL(k_trailing<32>, dim3(nrun * P.B), dim3(256), 0, s, K, strideK, P.ld, P.Wbuf, strideW, P.Npad,
                           P.T, kp, nsub, wslot, jlo, jhi, P.supertile, ntl, tpb, nrun, P.B, phase, want);
    else if (P.kc == 16)
        hipLaunchKernelGGL(k_colupdate<16>, dim3(nrun * P.B), dim3(256), 0, s, K, strideK, P.ld, P.Wbuf, strideW, P.Npad,
                           P.T, kp, nsub, wslot, jlo, jhi, P.supertile, ntl, tpb, nrun, P.B, phase, want);
    else
        hipLaunchKernelGGL(k_colupdate<32>, dim3(nrun * P.B), dim3(256), 0, s, K, strideK, P.ld, P.Wbuf, strideW, P.Npad,
                           P.T, kp, nsub, wslot, jlo, jhi, P.supertile, ntl, tpb, nrun, P.B, phase, want);
    if (timed) { hipEventRecord(ev.second, s); tm->pending_trailing.push_back(ev); }
    if (tm && count) tm->trailing_launches++;
}

static void launch_panel(const LdltPlan &P, hipStream_t s, double *K, double *dinv, int c, int wslot,
                         const int *phase, int want, double *b, double *v)
{
    const long strideK = (long)P.ld * P.Npad, strideW = (long)P.B * P.Npad * 64;
    hipLaunchKernelGGL(k_diag_factor, dim3(P.B), dim3(256), 0, s, K, strideK, P.ld, dinv, P.Npad, c, phase, want, b, v);
    const int rem = P.T - c - 1;
    if (rem <= 0) return;
    // one row tile per workgroup.  The TB = 2 instantiation (two row tiles share every L_kk read and every
    // barrier) is correct and spill-free but measured slower: 15.4 instead of 12.0 ms per factorisation, the
    // longer steps of half as many workgroups hide less latency.
    if (P.trsm_mfma)
        hipLaunchKernelGGL(k_panel_trsm_mfma, dim3((rem + 1) / 2, P.B), dim3(256), 0, s, K, strideK, P.ld, dinv,
                           P.Wbuf + (long)wslot * strideW, P.Npad, c, P.T, phase, want, b, c + 1, strideW,
                           (const unsigned char *)nullptr, 0);
    else
        hipLaunchKernelGGL(k_panel_trsm<1>, dim3(rem, P.B), dim3(256), 0, s, K, strideK, P.ld, dinv,
                           P.Wbuf + (long)wslot * strideW, P.Npad, c, P.T, phase, want, b);
}

// Two-level right-looking LDL^T: outer panels of R 64-wide sub-panels, so every pass over the
// trailing matrix applies a rank-64R update.  Look-ahead: the update is split into the "head" (the R
// tile columns of the next outer panel) and the "rest"; the latency-bound factorisation of the next
// panel runs on the auxiliary stream while the main stream streams the rest through the MFMA kernel.
// W = L D of the current outer panel lives in Wbuf slots [0,R) or [MAX_R, MAX_R+R) by panel parity.
// b, v (optional, [B][Npad]): the forward elimination L y = b is fused into the panel kernels (b <- y, v <- D^-1 y),
// so that the first solve after a factorisation is ldlt_solve(..., skip_fwd = true).
void ldlt_factor(const LdltPlan &P, double *K, double *dinv, const int *phase, int want, Timers *tm, double *b, double *v)
{
    const int T = P.T, R = P.R, Ts = P.Ts;
    hipStream_t sA = P.stream, sB = (P.aux && T - Ts >= P.lookahead_min) ? P.aux : P.stream;
    const long strideK = (long)P.ld * P.Npad, strideW = (long)P.B * P.Npad * 64;
    if (Ts > 0) {
        // ---- the Ts leading tile columns are mutually independent (order.hip): the tiles between them are zero and
        //      stay zero, so all diagonal tiles factor in one launch, all their panel solves (rows >= Ts only) in a
        //      second, and one rank-64 Ts update brings the dense remainder up to date
        hipLaunchKernelGGL(k_diag_factor, dim3(P.B, Ts), dim3(256), 0, sA, K, strideK, P.ld, dinv, P.Npad, 0, phase, want, b, v);
        if (T > Ts) {
            hipLaunchKernelGGL(k_panel_trsm_mfma, dim3((T - Ts + 1) / 2, P.B, Ts), dim3(256), 0, sA, K, strideK, P.ld,
                               dinv, P.Wbuf, P.Npad, 0, T, phase, want, (double *)nullptr, Ts, strideW, P.tmask, Ts);
            if (b)
                hipLaunchKernelGGL(k_fwd_lead, dim3(T - Ts, P.B), dim3(256), 0, sA, K, strideK, P.ld, b, P.Npad, Ts, phase, want,
                                   P.tmask);
            launch_update(P, sA, K, 0, Ts, 0, Ts, T, phase, want, tm, true, /*use_list=*/true);
        }
    }
    const int nq = (T - Ts + R - 1) / R;
    hipEvent_t evStart = P.ev[0];
    if (sB != sA) { hipEventRecord(evStart, sA); hipStreamWaitEvent(sB, evStart, 0); }
    for (int q = 0; q < nq; ++q) {
        const int c0 = Ts + R * q, slot = (q & 1) * LdltPlan::MAX_R;
        const int nsub = c0 + R <= T ? R : T - c0;
        hipEvent_t evPanel = P.ev[1 + (q & 1)], evHead = P.ev[3 + (q & 1)];
        // ---- stream B: factor the outer panel, sub-panel by sub-panel, left-looking inside the panel: tile
        //      column c0+j first receives the rank-64j update of sub-panels 0..j-1 in ONE pass (each in-panel
        //      column is read and written once; right-looking rank-64 updates cost twice the HBM traffic)
        for (int j = 0; j < nsub; ++j) {
            if (j) launch_update(P, sB, K, c0, j, slot, c0 + j, c0 + j + 1, phase, want, tm, false);
            launch_panel(P, sB, K, dinv, c0 + j, slot + j, phase, want, b, v);
        }
        if (c0 + nsub >= T) break;
        if (sB != sA) { hipEventRecord(evPanel, sB); hipStreamWaitEvent(sA, evPanel, 0); }
        // ---- stream A: head (the next panel's tile columns), then the rest
        launch_update(P, sA, K, c0, nsub, slot, c0 + R, c0 + 2 * R, phase, want, tm, true);
        if (sB != sA) { hipEventRecord(evHead, sA); hipStreamWaitEvent(sB, evHead, 0); }
        launch_update(P, sA, K, c0, nsub, slot, c0 + 2 * R, T, phase, want, tm, true);
    }
    if (sB != sA) { hipEventRecord(P.ev[5], sB); hipStreamWaitEvent(sA, P.ev[5], 0); }
    if (tm) tm->n_factor++;
}

// algorithmic flops of the k_trailing launches of one factorisation of one instance (what bench.py prices the
// kernel's event time with): per launch the lower triangle (incl. diagonal) of the updated block -- r (r + 1) / 2
// entries for r rows -- times 2 flops per multiply-add times the depth 64 nsub.  (The upper halves of the
// diagonal tiles, which the kernel also computes, are not counted.)
double ldlt_trailing_flops(const LdltPlan &P)
{
    const int T = P.T, R = P.R, Ts = P.Ts;
    double f = 0.0;
    auto tri = [](int tiles) { const double r = 64.0 * tiles; return r * (r + 1.0); };   // 2 * r (r + 1) / 2
    if (Ts > 0 && T > Ts) f += P.pair_ptr ? P.lead_update_flops : tri(T - Ts) * 64.0 * Ts;
    for (int c0 = Ts; c0 < T; c0 += R) {
        const int nsub = c0 + R <= T ? R : T - c0;
        if (c0 + nsub >= T) break;
        f += tri(T - c0 - nsub) * 64.0 * nsub;
    }
    return f;
}

void ldlt_solve(const LdltPlan &P, const double *K, const double *dinv, double *x, double *v,
                const int *phase, int want, bool skip_fwd)
{
    const long strideK = (long)P.ld * P.Npad;
    hipStream_t s = P.stream;
    const int Ts = P.Ts;
    if (!skip_fwd) {
        if (Ts > 0) {
            hipLaunchKernelGGL(k_fwd_step, dim3(1, P.B, Ts), dim3(64), 0, s, K, strideK, P.ld, dinv, x, v, P.Npad, 0,
                               phase, want, 1);
            if (P.T > Ts)
                hipLaunchKernelGGL(k_fwd_lead, dim3(P.T - Ts, P.B), dim3(256), 0, s, K, strideK, P.ld, x, P.Npad, Ts,
                                   phase, want, P.tmask);
        }
        for (int k = Ts; k < P.T; ++k)
            hipLaunchKernelGGL(k_fwd_step, dim3(P.T - k, P.B), dim3(64), 0, s, K, strideK, P.ld, dinv, x, v,
                               P.Npad, k, phase, want, 0);
    }
    for (int k = P.T - 1; k >= Ts; --k)
        hipLaunchKernelGGL(k_bwd_step, dim3(k + 1, P.B), dim3(256), 0, s, K, strideK, P.ld, x, v, P.Npad, k,
                           phase, want, 0, P.tmask, Ts);
    if (Ts > 0)
        hipLaunchKernelGGL(k_bwd_step, dim3(1, P.B, Ts), dim3(256), 0, s, K, strideK, P.ld, x, v, P.Npad, 0,
                           phase, want, 1, (const unsigned char *)nullptr, 0);
}

}  // namespace sqphip
