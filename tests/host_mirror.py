"""Stand-in for the reference's Julia host -- TEST HARNESS, not product.

The drop-in boundary of this project is the C ABI of include/sqphip.h, reached from Julia through julia/SqpHip.jl.
No Julia toolchain exists in the image, so the tests drive the seat the way the reference's host would: this file is a
line-by-line Python mirror of the host-side control flow that STAYS in Julia upstream

    Parameters      /root/reference/src/parameters.jl:1-30 (fields read by the hot path)
    Model           /root/reference/src/model.jl:3-68 (five callbacks + bounds + sparsity)
    SqpTR.run       /root/reference/src/algorithms/sqp_trust_region.jl:98-223, with every numerical step (sub-problem,
                    norms, KT residual, merit, q-model, ratio test) done by the library through
                    sqpsolver_jl_amd.Context / QpHip
    compute_step_Sl1QP   /root/reference/src/algorithms/sqp_trust_region.jl:393-471 ("not currently used" upstream)

It computes nothing numerical itself; it earns no coverage credit and is imported by tests/ and
__graft_entry__.smoke() only.
"""
from __future__ import annotations

import dataclasses
import math

import numpy as np

from sqpsolver_jl_amd.host import (Context, QpData, QpHip, default_options, _f, _OK, _INFEAS,  # noqa: F401
                                   MODE_QP, MODE_FR, MODE_SOC, MODE_LP, MODE_L1QP, MODE_INFEAS)


@dataclasses.dataclass
class Parameters:
    """src/parameters.jl:1-30 (fields read by the hot path)"""
    tol_direction: float = 1e-8
    tol_residual: float = 1e-8
    tol_infeas: float = 1e-8
    max_iter: int = 3000
    init_mu: float = 1.0
    tr_size: float = 10.0
    use_soc: bool = False


class Model:
    """src/model.jl:3-68: dimensions, bounds, sparsity (1-based COO) and the five callbacks.
    Callbacks: eval_f(x)->float, eval_grad_f(x)->grad, eval_g(x)->g, eval_jac_g(x)->vals,
    eval_h(x, obj_factor, lam)->vals (may be None)."""

    def __init__(self, n, m, x_L, x_U, g_L, g_U, j_str, h_str, eval_f, eval_g, eval_grad_f, eval_jac_g,
                 eval_h, num_linear_constraints, parameters: Parameters | None = None):
        self.n, self.m = n, m
        self.x = np.zeros(n)
        self.x_L, self.x_U, self.g_L, self.g_U = (_f(a) for a in (x_L, x_U, g_L, g_U))
        self.g = np.zeros(m)
        self.j_str, self.h_str = list(j_str), list(h_str)
        self.mult_g = np.zeros(m); self.mult_x_L = np.zeros(n); self.mult_x_U = np.zeros(n)
        self.obj_val = 0.0
        self.status = -5
        self.eval_f, self.eval_g, self.eval_grad_f = eval_f, eval_g, eval_grad_f
        self.eval_jac_g, self.eval_h = eval_jac_g, eval_h
        self.num_linear_constraints = num_linear_constraints
        self.parameters = parameters or Parameters()
        self.statistics = {}


def _isapprox(a, b):
    if a == b:
        return True
    if not (math.isfinite(a) and math.isfinite(b)):
        return False
    return abs(a - b) <= 1.4901161193847656e-08 * max(abs(a), abs(b))


class SqpTR:
    """sqp_trust_region.jl:6-91 state + run! (:98-223) on the host, every numerical step on the GPU."""

    def __init__(self, problem: Model):
        pr = problem
        self.problem = pr
        n, m = pr.n, pr.m
        self.x = pr.x.copy()
        self.p = np.zeros(n); self.p_soc = np.zeros(n)
        self.lam = np.zeros(m); self.mult_x_L = np.zeros(n); self.mult_x_U = np.zeros(n)
        self.df = np.zeros(n); self.E = np.zeros(m)
        self.dE = np.zeros(len(pr.j_str)); self.h_val = np.zeros(len(pr.h_str))
        self.f = 0.0
        self.phi = 1e20; self.mu = 1e4; self.Delta = 10.0; self.Delta_max = 1e8
        self.step_acceptance = True
        self.prim_infeas = math.inf; self.dual_infeas = math.inf
        self.feasibility_restoration = False
        self.iter = 1; self.ret = -5
        self.sub_status = None
        self.trace = []
        par = pr.parameters
        opts = default_options(tol_direction=par.tol_direction, tol_residual=par.tol_residual,
                               tol_infeas=par.tol_infeas, max_iter=par.max_iter, init_mu=par.init_mu,
                               tr_size=par.tr_size, use_soc=int(par.use_soc))
        jr = [r for r, _ in pr.j_str]; jc = [c for _, c in pr.j_str]
        hr = [r for r, _ in pr.h_str]; hc = [c for _, c in pr.h_str]
        self.ctx = Context(n, m, pr.num_linear_constraints, jr, jc, hr, hc, pr.x_L, pr.x_U, pr.g_L, pr.g_U,
                           opts, batch=1)
        self.optimizer = None

    # sqp.jl:86-117
    def eval_functions(self):
        pr = self.problem
        self.f = pr.eval_f(self.x)
        self.df = _f(pr.eval_grad_f(self.x))
        self.E = _f(pr.eval_g(self.x))
        self.dE = _f(pr.eval_jac_g(self.x))
        if pr.eval_h is not None:
            self.h_val = _f(pr.eval_h(self.x, 1.0, self.lam))

    def _qpdata(self, b=None):
        pr = self.problem
        return QpData(self.h_val if pr.eval_h is not None else None, self.df, self.dE,
                      self.E if b is None else b, pr.g_L, pr.g_U, pr.x_L, pr.x_U, pr.num_linear_constraints)

    def _push_trace(self):
        self.trace.append(dict(iter=self.iter, accepted=int(self.step_acceptance),
                               fr=int(self.feasibility_restoration), sub_status=self.sub_status, f=self.f,
                               phi=self.phi, mu=self.mu, delta=self.Delta,
                               pnorm=float(np.abs(self.p).max(initial=0.0)),
                               prim_infeas=self.prim_infeas, dual_infeas=self.dual_infeas))

    def compute_phi(self, x, alpha, p):     # sqp.jl:170-183
        pr = self.problem
        tmpx = x + alpha * p
        f, tmpE = self.f, self.E
        if alpha > 0.0:
            f = pr.eval_f(tmpx)
            tmpE = _f(pr.eval_g(tmpx))
        return self.ctx.compute_phi(f, tmpE, tmpx, self.mu, self.feasibility_restoration)

    def compute_qmodel(self, p, with_step):  # sqp_trust_region.jl:487-508
        return self.ctx.compute_qmodel(self.x, p, self.df, self.E, self.dE,
                                       self.h_val if self.problem.eval_h is not None else None, self.mu,
                                       with_step)

    def run(self):
        pr, par, ctx = self.problem, self.problem.parameters, self.ctx
        self.mu = par.init_mu
        self.Delta = par.tr_size
        # :237-254
        self.f = pr.eval_f(self.x)
        if not math.isnan(self.f):
            self.E = _f(pr.eval_g(self.x))
        lpviol = 0.0
        for i in range(pr.num_linear_constraints):
            lpviol += max(0.0, pr.g_L[i] - self.E[i]) - min(0.0, pr.g_U[i] - self.E[i])
        lpviol += float(np.maximum(0.0, pr.x_L - self.x).sum() - np.minimum(0.0, pr.x_U - self.x).sum())
        if math.isnan(self.f):
            pr.status = -13
            return
        if lpviol > par.tol_infeas:           # sub_optimize_lp! :264-304
            self.df = _f(pr.eval_grad_f(self.x))
            self.dE = _f(pr.eval_jac_g(self.x))
            qp = QpHip(ctx, self._qpdata())
            x, lam, mu_u, mu_l, st = qp.sub_optimize_lp(self.x)
            dz = lambda v: np.where(np.abs(v) < 1e-10, 0.0, v)
            self.x, self.lam, self.mult_x_U, self.mult_x_L, self.sub_status = dz(x), dz(lam), dz(mu_u), dz(mu_l), st
            self._push_trace()
        while True:
            if self.iter > par.max_iter:       # sqp.jl:215-224
                self.ret = 6 if self.prim_infeas <= par.tol_infeas else -1
                break
            if self.step_acceptance:           # :134-138
                self.eval_functions()
                self.prim_infeas = ctx.norm_violations(self.E, self.x, 1)
                self.dual_infeas = ctx.kt_residuals(self.df, self.lam, self.mult_x_U, self.mult_x_L, self.dE)
            # compute_step! :370-380
            if self.optimizer is None:
                self.optimizer = QpHip(ctx, self._qpdata())
                self.optimizer.create_model(self.Delta)
            else:
                self.optimizer.data = self._qpdata()
            if self.feasibility_restoration:
                out = self.optimizer.sub_optimize_FR(self.x, self.Delta)
            else:
                out = self.optimizer.sub_optimize(self.x, self.Delta)
            self.p, lam, mu_u, mu_l, _, self.sub_status = out
            p_lambda = lam - self.lam
            p_mult_x_L = mu_l - self.mult_x_L
            p_mult_x_U = mu_u - self.mult_x_U
            self.mu = max(self.mu, np.abs(self.lam).max(initial=0.0), np.abs(self.mult_x_L).max(initial=0.0),
                          np.abs(self.mult_x_U).max(initial=0.0))
            pn = float(np.abs(self.p).max(initial=0.0))
            if self.sub_status in _OK:
                if self.Delta == self.Delta_max and _isapprox(pn, self.Delta):
                    self.ret = 4
                    break
            elif self.sub_status in _INFEAS:
                if self.feasibility_restoration:
                    self.ret = 6 if self.prim_infeas <= par.tol_infeas else 2
                    break
                self.feasibility_restoration = True
                self._push_trace()
                self.iter += 1
                continue
            else:                               # quirk #1: ret stays -5 unless nearly feasible
                if self.prim_infeas <= par.tol_infeas * 10.0:
                    self.ret = 6
                break
            if self.step_acceptance:
                self.phi = self.compute_phi(self.x, 0.0, self.p)
            self._push_trace()
            if pn <= par.tol_direction:
                if self.feasibility_restoration:
                    self.feasibility_restoration = False
                    self.iter += 1
                    continue
                self.ret = 0
                break
            if (self.prim_infeas <= par.tol_infeas and self.dual_infeas <= par.tol_residual
                    and not _isapprox(self.Delta, pn) and not self.feasibility_restoration):
                self.ret = 0
                break
            # do_step! :515-579
            phi_k = self.compute_phi(self.x, 1.0, self.p)
            ared = self.phi - phi_k
            pred, q_0 = 1.0, 0.0
            if not self.feasibility_restoration:
                q_0 = self.compute_qmodel(self.p, False)
                pred = q_0 - self.compute_qmodel(self.p, True)
            accept, new_delta = ctx.tr_update(ared, pred, self.Delta, pn, self.Delta_max)
            if accept:
                self.x = self.x + self.p
                self.lam = self.lam + p_lambda
                self.mult_x_L = self.mult_x_L + p_mult_x_L
                self.mult_x_U = self.mult_x_U + p_mult_x_U
                self.Delta = new_delta
                self.step_acceptance = True
            else:
                perform_soc = False
                tmpx = self.x + self.p
                c_k = ctx.norm_violations(_f(pr.eval_g(tmpx)), tmpx, 1)
                if par.use_soc and c_k > 0 and not self.feasibility_restoration:
                    # sub_optimize_soc! :341-360 (E_soc = g(x+p) - J p, formed by the library's q-model path)
                    jp = self._jac_times(self.p)
                    e_soc = _f(pr.eval_g(tmpx)) - jp
                    self.optimizer.data = self._qpdata(b=e_soc)
                    r = ctx.qp_solve(MODE_SOC, self.x, self.Delta, self.mu, self.df, e_soc, self.dE,
                                     self.h_val if pr.eval_h is not None else None)
                    self.p_soc = self.p + r["p"]
                    phi_soc = self.compute_phi(self.x, 1.0, self.p_soc)
                    ared = self.phi - phi_soc
                    pred = q_0 - self.compute_qmodel(self.p_soc, True)
                    if ared > 0 and ared / pred > 0:
                        self.x = self.x + self.p_soc
                        self.lam = self.lam + p_lambda
                        self.mult_x_L = self.mult_x_L + p_mult_x_L
                        self.mult_x_U = self.mult_x_U + p_mult_x_U
                        self.step_acceptance = True
                        perform_soc = True
                if not perform_soc:
                    self.Delta = new_delta
                    self.step_acceptance = False
            if self.feasibility_restoration and self.step_acceptance:
                self.feasibility_restoration = False
            self.iter += 1
        # :215-222
        pr.obj_val = pr.eval_f(self.x)
        pr.status = int(self.ret)
        pr.x[:] = self.x
        pr.g[:] = self.E
        pr.mult_g[:] = -self.lam
        pr.mult_x_U[:] = -self.mult_x_U
        pr.mult_x_L[:] = self.mult_x_L
        pr.statistics["iter"] = self.iter

    def compute_step_Sl1QP(self, seat=None, max_mu=1e10, log=None):
        """sqp_trust_region.jl:393-471 ("not currently used" upstream): step from the elastic-mode QP with the penalty
        raised until the l1-QP is as feasible as the pure infeasibility problem allows.  `seat`: any object with the
        QpJuMP methods sub_optimize / sub_optimize_infeas / sub_optimize_L1QP (default: QpHip on this context) --
        the tests run the same driver over the device seat and over the oracle's.  The iterate must have been
        evaluated (eval_functions).  Returns (p, lambda, mult_x_U, mult_x_L, status)."""
        eps_1 = 0.9
        ctx = self.ctx
        if seat is None:
            seat = QpHip(ctx, self._qpdata())
        p, lam, mu_u, mu_l, slack, st = seat.sub_optimize(self.x, self.Delta)            # :398
        self.sub_status = st
        if st in (1, 10, 4):                       # OPTIMAL, ALMOST_LOCALLY_SOLVED, LOCALLY_SOLVED (:400)
            m_0 = ctx.norm_violations(self.E, self.x, 1)                                 # :402
            m_mu = float(np.sum(slack))                                                  # :403-406
            if m_mu > 1.0e-8:
                p_inf, infeasibility = seat.sub_optimize_infeas(self.x, self.Delta)      # :409
                if infeasibility < 1.0e-8:                                               # :411-421
                    while m_mu > 1.0e-8 and self.mu < max_mu:
                        self.mu = min(10.0 * self.mu, max_mu)
                        p, lam, mu_u, mu_l, slack, st = seat.sub_optimize_L1QP(self.x, self.Delta, self.mu)
                        self.sub_status = st
                        m_mu = float(np.sum(slack))
                        if log is not None:
                            log.append(("feasible", self.mu, st, m_mu))
                else:                                                                    # :422-441
                    m_inf = ctx.norm_violations(self.E + self._jac_times(p_inf), self.x + p_inf, 1)
                    while m_0 - m_mu < eps_1 * (m_0 - m_inf) and self.mu < max_mu:
                        self.mu = min(10.0 * self.mu, max_mu)
                        p, lam, mu_u, mu_l, slack, st = seat.sub_optimize_L1QP(self.x, self.Delta, self.mu)
                        self.sub_status = st
                        m_mu = float(np.sum(slack))
                        if log is not None:
                            log.append(("infeasible", self.mu, st, m_mu))
        self.p = p                                                                       # :464-468
        self.mu = max(self.mu, float(np.abs(self.lam).max(initial=0.0)))
        return p, lam, mu_u, mu_l, self.sub_status

    def _jac_times(self, p):
        """J p via two q-model evaluations is overkill; J is tiny on the host side of a drop-in, so
        the COO product is formed here (control-plane glue, O(nnzJ))."""
        out = np.zeros(self.problem.m)
        for k, (r, c) in enumerate(self.problem.j_str):
            out[r - 1] += self.dE[k] * p[c - 1]
        return out


def optimize(model: Model):
    """src/model.jl:70-91 for algorithm == "SQP-TR"."""
    sqp = SqpTR(model)
    sqp.run()
    return sqp
