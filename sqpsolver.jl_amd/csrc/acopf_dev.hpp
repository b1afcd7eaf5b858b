// acopf_dev.hpp -- device-side ACOPF evaluator (see acopf.hip for the seat in the reference)
#pragma once
#include "ctx.hpp"
#include "dev_util.hpp"
#include <cmath>

namespace sqphip {

// flow k of a branch (k = 0..3: p_f, q_f, p_t, q_t):  F_k = A v_self^2 + v_f v_t (Bc cos th + Bs sin th); the twelve
// coefficients per branch come from the host (tap ratio and phase shift folded in, acopf_synth.py branch_coeffs)
struct Ohm { double A, Bc, Bs; int self_t; };
static __device__ __forceinline__ Ohm ohm_coef(const double *__restrict__ oc, int k)
{
    return {oc[3 * k], oc[3 * k + 1], oc[3 * k + 2], k >= 2};
}

// Rectangular voltage coordinates (PowerModels ACRPowerModel under the build_opf of
// /root/reference/examples/acopf/opf.jl:12-43, the formulation run_sqp_opf instantiates at :46,:51), laid out by
// sqpsolver.jl_amd/acopf_synth.py acr_layout: x = (vi, vr, pg, qg, flows, dc lines); rows = vi[ref]; balance;
// vmin^2 <= vr^2 + vi^2 and vr^2 + vi^2 <= vmax^2 per bus; thermal limits; Ohm's law; dc-line losses.
//   F_k = A (vr_s^2 + vi_s^2) + Bc (vr_f vr_t + vi_f vi_t) + Bs (vi_f vr_t - vr_f vi_t)
// (the polar F_k with v_f v_t cos th and v_f v_t sin th written out, same twelve coefficients per branch): every row is
// quadratic, no trigonometry, the Hessian entries are multipliers times constants.
static __device__ __forceinline__ void acr_eval(const DV &d, int inst, const double *__restrict__ x, double sigma,
                         const double *__restrict__ lam, double *f_out, double *grad, double *gv,
                         double *jv, double *hv)
{
    const int nb = d.nb, ng = d.ng, nl = d.nl;
    const int VI = 0, VR = nb, PG = 2 * nb, PF = 2 * nb + 2 * ng;
    const int PT = PF + nl, QF = PF + 2 * nl, QT = PF + 3 * nl, DCV = PF + 4 * nl;
    const int V0 = 1 + 2 * nb, T0 = V0 + 2 * nb, O0 = T0 + 2 * nl, D0 = O0 + 4 * nl;
    const int nbal = d.bal_ptr[nb];
    const int JV = 1 + 2 * nbal, JT = JV + 4 * nb, JO = JT + 4 * nl, JS = JO + 20 * nl, JD = JS + 4 * d.nsh;
    const int HT = ng, HV = ng + 4 * nl, HO = HV + 4 * nb, HS = HO + 24 * nl;
    const double *ohm = d.br_ohm + (long)inst * nl * 12;
    const double *c2 = d.c2 + (long)inst * ng, *c1 = d.c1 + (long)inst * ng;
    if (f_out) {
        double f = 0.0;
        for (int g = threadIdx.x; g < ng; g += TPB) f += c2[g] * x[PG + g] * x[PG + g] + c1[g] * x[PG + g];
        f = block_reduce<OpSum>(f);
        if (threadIdx.x == 0) *f_out = f;
    }
    if (grad) {
        for (int j = threadIdx.x; j < d.n; j += TPB) grad[j] = 0.0;
        __syncthreads();
        for (int g = threadIdx.x; g < ng; g += TPB) grad[PG + g] = 2 * c2[g] * x[PG + g] + c1[g];
    }
    if (hv) {
        for (int g = threadIdx.x; g < ng; g += TPB) hv[g] = sigma * 2 * c2[g];
    }
    if (threadIdx.x == 0) {
        if (gv) gv[0] = x[VI + d.ref_bus];
        if (jv) jv[0] = 1.0;
    }
    // bus rows: balance (+ shunt), voltage magnitude
    for (int i = threadIdx.x; i < nb; i += TPB) {
        const double vr = x[VR + i], vi = x[VI + i], w2 = vr * vr + vi * vi;
        const int s = d.bal_ptr[i], e = d.bal_ptr[i + 1];
        if (gv) {
            double sp = 0.0, sq = 0.0;
            for (int k = s; k < e; ++k) {
                sp += d.bal_coef[k] * x[d.bal_colP[k]];
                sq += d.bal_coef[k] * x[d.bal_colQ[k]];
            }
            if (d.nsh > 0 && d.sh_of_bus[i] >= 0) {
                const int sh = d.sh_of_bus[i];
                sp += d.sh_gs[sh] * w2; sq -= d.sh_bs[sh] * w2;
            }
            gv[1 + 2 * i] = sp; gv[2 + 2 * i] = sq;
            gv[V0 + 2 * i] = w2; gv[V0 + 2 * i + 1] = w2;
        }
        if (jv) {
            double *dst = jv + 1 + 2 * s;
            for (int k = s; k < e; ++k) { dst[k - s] = d.bal_coef[k]; dst[(e - s) + (k - s)] = d.bal_coef[k]; }
            jv[JV + 2 * i] = 2 * vr; jv[JV + 2 * i + 1] = 2 * vi;
            jv[JV + 2 * nb + 2 * i] = 2 * vr; jv[JV + 2 * nb + 2 * i + 1] = 2 * vi;
        }
        if (hv) {
            const double wl = 2 * lam[V0 + 2 * i], wu = 2 * lam[V0 + 2 * i + 1];
            hv[HV + 4 * i] = wl; hv[HV + 4 * i + 1] = wl; hv[HV + 4 * i + 2] = wu; hv[HV + 4 * i + 3] = wu;
        }
    }
    for (int sh = threadIdx.x; sh < d.nsh; sh += TPB) {
        const int i = d.sh_bus[sh];
        if (jv) {
            jv[JS + 4 * sh] = 2 * d.sh_gs[sh] * x[VR + i]; jv[JS + 4 * sh + 1] = 2 * d.sh_gs[sh] * x[VI + i];
            jv[JS + 4 * sh + 2] = -2 * d.sh_bs[sh] * x[VR + i]; jv[JS + 4 * sh + 3] = -2 * d.sh_bs[sh] * x[VI + i];
        }
        if (hv) {
            const double w = lam[1 + 2 * i] * 2 * d.sh_gs[sh] - lam[2 + 2 * i] * 2 * d.sh_bs[sh];
            hv[HS + 2 * sh] = w; hv[HS + 2 * sh + 1] = w;
        }
    }
    for (int dl = threadIdx.x; dl < d.ndc; dl += TPB) {
        if (gv) gv[D0 + dl] = (1.0 - d.dc_loss1[dl]) * x[DCV + dl] + x[DCV + d.ndc + dl];
        if (jv) { jv[JD + 2 * dl] = 1.0 - d.dc_loss1[dl]; jv[JD + 2 * dl + 1] = 1.0; }
    }
    // branch rows
    for (int l = threadIdx.x; l < nl; l += TPB) {
        const int fb = d.f_bus[l], tb = d.t_bus[l];
        const double vrf = x[VR + fb], vif = x[VI + fb], vrt = x[VR + tb], vit = x[VI + tb];
        const double pf = x[PF + l], qf = x[QF + l], pt = x[PT + l], qt = x[QT + l];
        const double cc = vrf * vrt + vif * vit, ss = vif * vrt - vrf * vit;
        if (gv) {
            gv[T0 + 2 * l] = pf * pf + qf * qf;
            gv[T0 + 2 * l + 1] = pt * pt + qt * qt;
        }
        if (jv) {
            jv[JT + 2 * l] = 2 * pf; jv[JT + 2 * l + 1] = 2 * qf;
            jv[JT + 2 * nl + 2 * l] = 2 * pt; jv[JT + 2 * nl + 2 * l + 1] = 2 * qt;
        }
        if (hv) {
            const double wf = 2 * lam[T0 + 2 * l], wt = 2 * lam[T0 + 2 * l + 1];
            hv[HT + 2 * l] = wf; hv[HT + 2 * l + 1] = wf;
            hv[HT + 2 * nl + 2 * l] = wt; hv[HT + 2 * nl + 2 * l + 1] = wt;
        }
        const double own[4] = {pf, qf, pt, qt};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const Ohm o = ohm_coef(ohm + 12 * l, k);
            if (gv) {
                const double w2 = o.self_t ? vrt * vrt + vit * vit : vrf * vrf + vif * vif;
                gv[O0 + 4 * l + k] = own[k] - (o.A * w2 + o.Bc * cc + o.Bs * ss);
            }
            if (jv) {
                double *e = jv + JO + (long)k * 5 * nl + 5 * l;
                e[0] = 1.0;
                e[1] = -((o.self_t ? 0.0 : 2 * o.A * vif) + o.Bc * vit + o.Bs * vrt);
                e[2] = -((o.self_t ? 2 * o.A * vit : 0.0) + o.Bc * vif - o.Bs * vrf);
                e[3] = -((o.self_t ? 0.0 : 2 * o.A * vrf) + o.Bc * vrt - o.Bs * vit);
                e[4] = -((o.self_t ? 2 * o.A * vrt : 0.0) + o.Bc * vrf + o.Bs * vif);
            }
            if (hv) {
                double *blk = hv + HO + (long)k * 6 * nl;
                const double w = -lam[O0 + 4 * l + k];
                blk[0 * nl + l] = w * 2 * o.A;
                blk[1 * nl + l] = w * 2 * o.A;
                blk[2 * nl + l] = w * o.Bc;
                blk[3 * nl + l] = w * o.Bc;
                blk[4 * nl + l] = w * o.Bs;
                blk[5 * nl + l] = -w * o.Bs;
            }
        }
    }
}

// W-space form (/root/reference/examples/acopf/acwr.jl:1-37 over PowerModels' build_opf; layout: acopf_synth.py
// acwr_layout): x = (vi, vr, w, wr, wi, pg, qg, flows, dc lines).  Balance, angle-difference and Ohm rows are linear in
// (w, wr, wi) -- their Jacobian entries are constants of the instance --, constraint_model_voltage ties the lifted
// variables to the rectangular voltages: w_i = vr_i^2 + vi_i^2, wr_k = vr_i vr_j + vi_i vi_j, wi_k = vi_i vr_j - vr_i vi_j.
static __device__ __forceinline__ void acwr_eval(const DV &d, int inst, const double *__restrict__ x, double sigma,
                          const double *__restrict__ lam, double *f_out, double *grad, double *gv,
                          double *jv, double *hv)
{
    const int nb = d.nb, ng = d.ng, nl = d.nl, nbp = d.nbp;
    const int VI = 0, VR = nb, W = 2 * nb, WR = 3 * nb, WI = 3 * nb + nbp, PG = 3 * nb + 2 * nbp, PF = PG + 2 * ng;
    const int PT = PF + nl, QF = PF + 2 * nl, QT = PF + 3 * nl, DCV = PF + 4 * nl;
    const int A0 = 1 + 2 * nb, O0 = A0 + 2 * nbp, V0 = O0 + 4 * nl, T0 = V0 + nb + 2 * nbp, D0 = T0 + 2 * nl;
    const int nbal = d.bal_ptr[nb];
    const int JS = 1 + 2 * nbal, JA = JS + 2 * nb, JA2 = JA + 2 * nbp, JO = JA2 + 2 * nbp, JW = JO + 16 * nl,
              JWR = JW + 3 * nb, JWI = JWR + 5 * nbp, JT = JWI + 5 * nbp, JD = JT + 4 * nl;
    const int HT = ng, HW = ng + 4 * nl, HWR = HW + 2 * nb, HWI = HWR + 2 * nbp;
    const double *ohm = d.br_ohm + (long)inst * nl * 12;
    const double *c2 = d.c2 + (long)inst * ng, *c1 = d.c1 + (long)inst * ng;
    if (f_out) {
        double f = 0.0;
        for (int g = threadIdx.x; g < ng; g += TPB) f += c2[g] * x[PG + g] * x[PG + g] + c1[g] * x[PG + g];
        f = block_reduce<OpSum>(f);
        if (threadIdx.x == 0) *f_out = f;
    }
    if (grad) {
        for (int j = threadIdx.x; j < d.n; j += TPB) grad[j] = 0.0;
        __syncthreads();
        for (int g = threadIdx.x; g < ng; g += TPB) grad[PG + g] = 2 * c2[g] * x[PG + g] + c1[g];
    }
    if (hv) {
        for (int g = threadIdx.x; g < ng; g += TPB) hv[g] = sigma * 2 * c2[g];
    }
    if (threadIdx.x == 0) {
        if (gv) gv[0] = x[VI + d.ref_bus];
        if (jv) jv[0] = 1.0;
    }
    for (int i = threadIdx.x; i < nb; i += TPB) {
        const int s = d.bal_ptr[i], e = d.bal_ptr[i + 1];
        const int sh = d.nsh > 0 ? d.sh_of_bus[i] : -1;
        const double gs = sh >= 0 ? d.sh_gs[sh] : 0.0, bs = sh >= 0 ? d.sh_bs[sh] : 0.0;
        const double vr = x[VR + i], vi = x[VI + i];
        if (gv) {
            double sp = 0.0, sq = 0.0;
            for (int k = s; k < e; ++k) {
                sp += d.bal_coef[k] * x[d.bal_colP[k]];
                sq += d.bal_coef[k] * x[d.bal_colQ[k]];
            }
            gv[1 + 2 * i] = sp + gs * x[W + i]; gv[2 + 2 * i] = sq - bs * x[W + i];
            gv[V0 + i] = x[W + i] - vr * vr - vi * vi;
        }
        if (jv) {
            double *dst = jv + 1 + 2 * s;
            for (int k = s; k < e; ++k) { dst[k - s] = d.bal_coef[k]; dst[(e - s) + (k - s)] = d.bal_coef[k]; }
            jv[JS + 2 * i] = gs; jv[JS + 2 * i + 1] = -bs;
            jv[JW + 3 * i] = 1.0; jv[JW + 3 * i + 1] = -2 * vr; jv[JW + 3 * i + 2] = -2 * vi;
        }
        if (hv) { const double w = -2 * lam[V0 + i]; hv[HW + 2 * i] = w; hv[HW + 2 * i + 1] = w; }
    }
    for (int k = threadIdx.x; k < nbp; k += TPB) {
        const int i = d.bp_i[k], j = d.bp_j[k];
        const double vri = x[VR + i], vii = x[VI + i], vrj = x[VR + j], vij = x[VI + j];
        if (gv) {
            gv[A0 + 2 * k] = x[WI + k] - d.bp_tmax[k] * x[WR + k];
            gv[A0 + 2 * k + 1] = x[WI + k] - d.bp_tmin[k] * x[WR + k];
            gv[V0 + nb + 2 * k] = x[WR + k] - (vri * vrj + vii * vij);
            gv[V0 + nb + 2 * k + 1] = x[WI + k] - (vii * vrj - vri * vij);
        }
        if (jv) {
            jv[JA + 2 * k] = 1.0; jv[JA + 2 * k + 1] = -d.bp_tmax[k];
            jv[JA2 + 2 * k] = 1.0; jv[JA2 + 2 * k + 1] = -d.bp_tmin[k];
            double *a = jv + JWR + 5 * k, *b = jv + JWI + 5 * k;
            a[0] = 1.0; a[1] = -vrj; a[2] = -vri; a[3] = -vij; a[4] = -vii;
            b[0] = 1.0; b[1] = -vrj; b[2] = -vii; b[3] = vij; b[4] = vri;
        }
        if (hv) {
            const double wr = -lam[V0 + nb + 2 * k], wi = lam[V0 + nb + 2 * k + 1];
            hv[HWR + 2 * k] = wr; hv[HWR + 2 * k + 1] = wr;
            hv[HWI + 2 * k] = -wi; hv[HWI + 2 * k + 1] = wi;
        }
    }
    for (int dl = threadIdx.x; dl < d.ndc; dl += TPB) {
        if (gv) gv[D0 + dl] = (1.0 - d.dc_loss1[dl]) * x[DCV + dl] + x[DCV + d.ndc + dl];
        if (jv) { jv[JD + 2 * dl] = 1.0 - d.dc_loss1[dl]; jv[JD + 2 * dl + 1] = 1.0; }
    }
    for (int l = threadIdx.x; l < nl; l += TPB) {
        const double pf = x[PF + l], qf = x[QF + l], pt = x[PT + l], qt = x[QT + l];
        const int k = d.br_bp[l];
        const double sg = d.br_sig[l], wrk = x[WR + k], wik = x[WI + k];
        const double wf = x[W + d.f_bus[l]], wt = x[W + d.t_bus[l]];
        if (gv) {
            gv[T0 + 2 * l] = pf * pf + qf * qf;
            gv[T0 + 2 * l + 1] = pt * pt + qt * qt;
        }
        if (jv) {
            jv[JT + 2 * l] = 2 * pf; jv[JT + 2 * l + 1] = 2 * qf;
            jv[JT + 2 * nl + 2 * l] = 2 * pt; jv[JT + 2 * nl + 2 * l + 1] = 2 * qt;
        }
        if (hv) {
            const double hf = 2 * lam[T0 + 2 * l], ht = 2 * lam[T0 + 2 * l + 1];
            hv[HT + 2 * l] = hf; hv[HT + 2 * l + 1] = hf;
            hv[HT + 2 * nl + 2 * l] = ht; hv[HT + 2 * nl + 2 * l + 1] = ht;
        }
        const double own[4] = {pf, qf, pt, qt};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const Ohm o = ohm_coef(ohm + 12 * l, c);
            if (gv) gv[O0 + 4 * l + c] = own[c] - (o.A * (o.self_t ? wt : wf) + o.Bc * wrk + sg * o.Bs * wik);
            if (jv) {
                double *e = jv + JO + (long)c * 4 * nl + 4 * l;
                e[0] = 1.0; e[1] = -o.A; e[2] = -o.Bc; e[3] = -sg * o.Bs;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// A synthetic NLP with a DENSE Lagrangian Hessian (bench.py --workload dense; BASELINE.json north_star: "dense LDL^T tiled
// on MFMA where the Hessian is dense").  No reference counterpart -- the reference's examples are all ACOPF -- but the
// shape SqpSolver.Model describes (src/model.jl:3-35): callbacks f, grad f, g, Jacobian, Hessian of the Lagrangian.
//     min  1/2 x'Qx + c'x + kappa/4 sum_i x_i^4     s.t.  A x = b (m linear rows),  xL <= x <= xU
// Q (n x n, symmetric, dense; shared by the batch) and A (m x n, dense; shared) live in HBM once; an instance carries
// c[n] and its bounds / right-hand sides.  Structure: Jacobian COO = A row-major, Hessian COO = lower triangle of Q
// column-major (sqpsolver.jl_amd/dense_synth.py).  Hessian of the Lagrangian: sigma (Q + 3 kappa diag(x^2)) -- the rows
// are linear.
static __device__ __forceinline__ void dense_eval(const DV &d, int inst, const double *__restrict__ x, double sigma,
                           const double *__restrict__ lam, double *f_out, double *grad, double *gv,
                           double *jv, double *hv)
{
    (void)lam;
    const int n = d.n, m = d.m;
    const double *Q = d.dnQ, *A = d.dnA, *c = d.dnc + (long)inst * n;
    const double kap = d.dn_kappa;
    if (f_out || grad) {
        // (Q x)_j: thread j walks column j of the symmetric Q (coalesced across the threads of a wave: consecutive j)
        double f = 0.0;
        for (int j = threadIdx.x; j < n; j += TPB) {
            double acc = 0.0;
            for (int i = 0; i < n; ++i) acc += Q[(long)i * n + j] * x[i];
            const double xj = x[j], x2 = xj * xj;
            if (grad) grad[j] = acc + c[j] + kap * x2 * xj;
            f += 0.5 * xj * acc + c[j] * xj + 0.25 * kap * x2 * x2;
        }
        if (f_out) {
            f = block_reduce<OpSum>(f);
            if (threadIdx.x == 0) *f_out = f;
        }
    }
    if (gv)
        for (int i = threadIdx.x; i < m; i += TPB) {
            double acc = 0.0;
            for (int j = 0; j < n; ++j) acc += A[(long)i * n + j] * x[j];
            gv[i] = acc;
        }
    if (jv) for (long k = threadIdx.x; k < (long)m * n; k += TPB) jv[k] = A[k];
    if (hv)
        for (int j = 0; j < n; ++j) {                       // column j of the lower triangle: entries i = j .. n - 1 at off(j) + i - j
            const long off = (long)j * n - (long)j * (j - 1) / 2;
            for (int i = j + threadIdx.x; i < n; i += TPB)
                hv[off + i - j] = sigma * (Q[(long)j * n + i] + (i == j ? 3.0 * kap * x[j] * x[j] : 0.0));
        }
}

// any of f_out, grad, gv, jv, hv may be null
static __device__ __forceinline__ void acopf_eval(const DV &d, int inst, const double *__restrict__ x, double sigma,
                           const double *__restrict__ lam, double *f_out, double *grad, double *gv,
                           double *jv, double *hv)
{
    if (d.dense_nlp) { dense_eval(d, inst, x, sigma, lam, f_out, grad, gv, jv, hv); return; }   // uniform over the launch
    if (d.acr) { acr_eval(d, inst, x, sigma, lam, f_out, grad, gv, jv, hv); return; }   // uniform over the launch
    if (d.acwr) { acwr_eval(d, inst, x, sigma, lam, f_out, grad, gv, jv, hv); return; }
    const int nb = d.nb, ng = d.ng, nl = d.nl;
    const int VA = 0, VM = nb, PG = 2 * nb, QG = 2 * nb + ng, PF = 2 * nb + 2 * ng;
    const int PT = PF + nl, QF = PF + 2 * nl, QT = PF + 3 * nl;
    const int T0 = 2 * nl + 1 + 2 * nb, O0 = T0 + 2 * nl;
    const double *ohm = d.br_ohm + (long)inst * nl * 12;
    const double *c2 = d.c2 + (long)inst * ng, *c1 = d.c1 + (long)inst * ng;
    (void)QG;
    // objective and gradient
    if (f_out) {
        double f = 0.0;
        for (int g = threadIdx.x; g < ng; g += TPB) f += c2[g] * x[PG + g] * x[PG + g] + c1[g] * x[PG + g];
        f = block_reduce<OpSum>(f);
        if (threadIdx.x == 0) *f_out = f;
    }
    if (grad) {
        for (int j = threadIdx.x; j < d.n; j += TPB) grad[j] = 0.0;
        __syncthreads();
        for (int g = threadIdx.x; g < ng; g += TPB) grad[PG + g] = 2 * c2[g] * x[PG + g] + c1[g];
    }
    if (hv) {
        for (int g = threadIdx.x; g < ng; g += TPB) hv[g] = sigma * 2 * c2[g];
    }
    // bus rows
    if (gv) {
        if (threadIdx.x == 0) gv[2 * nl] = x[VA + d.ref_bus];
        for (int i = threadIdx.x; i < nb; i += TPB) {
            double sp = 0.0, sq = 0.0;
            for (int k = d.bal_ptr[i]; k < d.bal_ptr[i + 1]; ++k) {
                sp += d.bal_coef[k] * x[d.bal_colP[k]];
                sq += d.bal_coef[k] * x[d.bal_colQ[k]];
            }
            if (d.nsh > 0 && d.sh_of_bus[i] >= 0) {          // bus shunt: + gs vm^2 (P), - bs vm^2 (Q)
                const int s = d.sh_of_bus[i];
                const double vm = x[VM + i];
                sp += d.sh_gs[s] * vm * vm; sq -= d.sh_bs[s] * vm * vm;
            }
            gv[2 * nl + 1 + 2 * i] = sp; gv[2 * nl + 2 + 2 * i] = sq;
        }
    }
    if (jv) {
        if (threadIdx.x == 0) jv[4 * nl] = 1.0;
        const int B0 = 4 * nl + 1;
        for (int i = threadIdx.x; i < nb; i += TPB) {
            const int s = d.bal_ptr[i], e = d.bal_ptr[i + 1];
            double *dst = jv + B0 + 2 * s;
            for (int k = s; k < e; ++k) { dst[k - s] = d.bal_coef[k]; dst[(e - s) + (k - s)] = d.bal_coef[k]; }
        }
    }
    // branch rows
    const int TH = 4 * nl + 1 + 2 * d.bal_ptr[nb], OH = TH + 4 * nl;
    const int HO = ng + 4 * nl;
    // HVDC lines: loss row d (behind every other row) = (1 - loss1) p_dc_f + p_dc_t; Jacobian entries behind the
    // shunt entries; no second derivatives
    for (int dl = threadIdx.x; dl < d.ndc; dl += TPB) {
        const int DCV = PF + 4 * nl;
        if (gv) gv[O0 + 4 * nl + dl] = (1.0 - d.dc_loss1[dl]) * x[DCV + dl] + x[DCV + d.ndc + dl];
        if (jv) { jv[OH + 20 * nl + 2 * d.nsh + 2 * dl] = 1.0 - d.dc_loss1[dl]; jv[OH + 20 * nl + 2 * d.nsh + 2 * dl + 1] = 1.0; }
    }
    // shunt entries at the end of both COO lists
    for (int s = threadIdx.x; s < d.nsh; s += TPB) {
        const int i = d.sh_bus[s];
        const double vm = x[VM + i];
        if (jv) { jv[OH + 20 * nl + 2 * s] = 2 * d.sh_gs[s] * vm; jv[OH + 20 * nl + 2 * s + 1] = -2 * d.sh_bs[s] * vm; }
        if (hv) hv[HO + 40 * nl + s] = lam[2 * nl + 1 + 2 * i] * 2 * d.sh_gs[s] - lam[2 * nl + 2 + 2 * i] * 2 * d.sh_bs[s];
    }
    for (int l = threadIdx.x; l < nl; l += TPB) {
        const int fb = d.f_bus[l], tb = d.t_bus[l];
        const double th = x[VA + fb] - x[VA + tb];
        const double vf = x[VM + fb], vt = x[VM + tb];
        const double pf = x[PF + l], qf = x[QF + l], pt = x[PT + l], qt = x[QT + l];
        double sn, cs;
        sincos(th, &sn, &cs);
        const double uu = vf * vt;
        if (gv) {
            gv[l] = th; gv[nl + l] = th;
            gv[T0 + 2 * l] = pf * pf + qf * qf;
            gv[T0 + 2 * l + 1] = pt * pt + qt * qt;
        }
        if (jv) {
            jv[2 * l] = 1.0; jv[2 * l + 1] = -1.0;
            jv[2 * nl + 2 * l] = 1.0; jv[2 * nl + 2 * l + 1] = -1.0;
            jv[TH + 2 * l] = 2 * pf; jv[TH + 2 * l + 1] = 2 * qf;
            jv[TH + 2 * nl + 2 * l] = 2 * pt; jv[TH + 2 * nl + 2 * l + 1] = 2 * qt;
        }
        if (hv) {
            const double wf = 2 * lam[T0 + 2 * l], wt = 2 * lam[T0 + 2 * l + 1];
            hv[ng + 2 * l] = wf; hv[ng + 2 * l + 1] = wf;
            hv[ng + 2 * nl + 2 * l] = wt; hv[ng + 2 * nl + 2 * l + 1] = wt;
        }
        const double own[4] = {pf, qf, pt, qt};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const Ohm o = ohm_coef(ohm + 12 * l, k);
            const double T0v = o.Bc * cs + o.Bs * sn, T1 = -o.Bc * sn + o.Bs * cs;
            if (gv) {
                const double vs = o.self_t ? vt : vf;
                gv[O0 + 4 * l + k] = own[k] - (o.A * vs * vs + uu * T0v);
            }
            if (jv) {
                double *e = jv + OH + (long)k * 5 * nl + 5 * l;
                e[0] = 1.0;
                e[1] = -(uu * T1);
                e[2] = uu * T1;
                e[3] = -((o.self_t ? 0.0 : 2 * o.A * vf) + vt * T0v);
                e[4] = -((o.self_t ? 2 * o.A * vt : 0.0) + vf * T0v);
            }
            if (hv) {
                double *blk = hv + HO + (long)k * 10 * nl;
                const double w = -lam[O0 + 4 * l + k];
                blk[0 * nl + l] = w * (-uu * T0v);
                blk[1 * nl + l] = w * (uu * T0v);
                blk[2 * nl + l] = w * (-uu * T0v);
                blk[3 * nl + l] = w * (vt * T1);
                blk[4 * nl + l] = w * (-vt * T1);
                blk[5 * nl + l] = w * (o.self_t ? 0.0 : 2 * o.A);
                blk[6 * nl + l] = w * (vf * T1);
                blk[7 * nl + l] = w * (-vf * T1);
                blk[8 * nl + l] = w * T0v;
                blk[9 * nl + l] = w * (o.self_t ? 2 * o.A : 0.0);
            }
        }
    }
}

}  // namespace sqphip
