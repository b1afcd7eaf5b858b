"""Host-side binding of the libsqphip C ABI with the reference's operator interface for the replaced path.

The reference host is Julia (`SqpSolver.Optimizer` -> `Model` -> `SqpTR.run!`); it stays Julia and reaches the
library through the `ccall` shim of julia/SqpHip.jl.  No Julia toolchain exists in this image, so
the same thin layer is written here in Python over ctypes with the reference's names, argument meaning and error
behaviour:

    Context         one sqphip_ctx: every entry point of include/sqphip.h
    QpData          /root/reference/src/algorithms/subproblem.jl:12-23
    QpHip           the `AbstractSubOptimizer` seat (subproblem.jl:1), API of QpJuMP:
                    sub_optimize / sub_optimize_FR / sub_optimize_lp / sub_optimize_L1QP /
                    sub_optimize_infeas  (subproblem_JuMP.jl:127-183, :352-393, :185-244, :283-347, :398-429)

The stand-in for the Julia HOST itself (`Model`, `Parameters`, `SqpTR.run!`: a transliteration of
sqp_trust_region.jl:98-223 that calls the seat) is test harness, not product, and lives in tests/host_mirror.py.

Nothing here computes on the CPU beyond control flow; if libsqphip.so is missing every entry point
raises.  The oracle under oracle/ is never imported from this package.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
import math
import numpy as np

from . import _lib

MODE_QP, MODE_FR, MODE_SOC, MODE_LP, MODE_L1QP, MODE_INFEAS = range(6)
# MOI.TerminationStatusCode integers
LOCALLY_SOLVED, LOCALLY_INFEASIBLE, ITERATION_LIMIT, NUMERICAL_ERROR = 4, 5, 11, 20
_OK = (1, 7, 10, 4)          # OPTIMAL, ALMOST_OPTIMAL, ALMOST_LOCALLY_SOLVED, LOCALLY_SOLVED
_INFEAS = (2, 5)             # INFEASIBLE, LOCALLY_INFEASIBLE

_dp = C.POINTER(C.c_double)


def _d(a):
    return None if a is None else a.ctypes.data_as(_dp)


def _f(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _l(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class SqpHipError(RuntimeError):
    pass


def default_options(**kw) -> _lib.Options:
    o = _lib.Options()
    _lib.lib().sqphip_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise KeyError(k)
        setattr(o, k, v)
    return o


def kkt_order(n, m, jrow, jcol, hrow, hcol, gL, gU, rows_last=True):
    """(pos, n_lead_tiles, order) of `sqphip_kkt_order`: host-only, works without a GPU."""
    L = _lib.lib()
    jr, jc, hr, hc = (np.ascontiguousarray(a, dtype=np.int64) for a in (jrow, jcol, hrow, hcol))
    # rows that stay in the condensed matrix: the equalities and rows with more than 32 entries (csrc/sparse.hpp)
    mk = int(np.sum((np.asarray(gL) == np.asarray(gU)) | (np.bincount(jr - 1, minlength=m) > 32)))
    pos = np.zeros(n + mk, dtype=np.int32); ts = C.c_int32(); nf = C.c_int32()
    rc = L.sqphip_kkt_order(n, m, len(jr), jr.ctypes.data_as(C.POINTER(C.c_int64)), jc.ctypes.data_as(C.POINTER(C.c_int64)),
                            len(hr), hr.ctypes.data_as(C.POINTER(C.c_int64)), hc.ctypes.data_as(C.POINTER(C.c_int64)),
                            _d(_f(gL)), _d(_f(gU)), int(rows_last), pos.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(ts), C.byref(nf))
    if rc != 0:
        raise SqpHipError(f"sqphip_kkt_order failed ({rc})")
    return pos, ts.value, nf.value


def kkt_symbolic(n, m, jrow, jcol, hrow, hcol, gL, gU, condense=True, rows_after_vars=True, small_front=0,
                 zero_frac=-1.0):
    """(pos, stats dict) of `sqphip_kkt_symbolic`: the symbolic analysis of the sparse Newton matrix; host-only."""
    L = _lib.lib()
    jr, jc, hr, hc = (np.ascontiguousarray(a, dtype=np.int64) for a in (jrow, jcol, hrow, hcol))
    pos = np.zeros(n + m, dtype=np.int32); st = _lib.SymbolicStats()
    rc = L.sqphip_kkt_symbolic(n, m, len(jr), _l(jr), _l(jc), len(hr), _l(hr), _l(hc), _d(_f(gL)), _d(_f(gU)),
                               int(condense), int(rows_after_vars), int(small_front), float(zero_frac), _i(pos), C.byref(st))
    if rc != 0:
        raise SqpHipError(f"sqphip_kkt_symbolic failed ({rc})")
    out = {k: getattr(st, k) for k, _ in _lib.SymbolicStats._fields_}
    return pos[:out["order"]], out


def mf_host_solve(n, m, jrow, jcol, hrow, hcol, gL, gU, condense, jval, hval, Dd, sigp, hd, rtype, hsc, dw, rhs):
    """Host reference of the multifrontal numeric phase (`sqphip_mf_host_solve`): (sol, dinv_by_unknown, npos)."""
    L = _lib.lib()
    jr, jc, hr, hc = (np.ascontiguousarray(a, dtype=np.int64) for a in (jrow, jcol, hrow, hcol))
    rhs = _f(rhs); sol = np.zeros_like(rhs); dinv = np.zeros_like(rhs); npos = C.c_int32()
    rt = np.ascontiguousarray(rtype, dtype=np.int32)
    rc = L.sqphip_mf_host_solve(n, m, len(jr), _l(jr), _l(jc), len(hr), _l(hr), _l(hc), _d(_f(gL)), _d(_f(gU)),
                                int(condense), _d(_f(jval)), _d(_f(hval)), _d(_f(Dd)), _d(_f(sigp)), _d(_f(hd)), _i(rt),
                                float(hsc), float(dw), _d(rhs), _d(sol), _d(dinv), C.byref(npos))
    if rc != 0:
        raise SqpHipError(f"sqphip_mf_host_solve failed ({rc})")
    return sol, dinv, npos.value


def mf_host_top2_err() -> float:
    """After `mf_host_solve`: relative error of the host replay of the streamed top-of-tree solve from its own plan arrays
    against the plain recursion of that call (-1: the plan has no such top)."""
    return float(_lib.lib().sqphip_mf_host_top2_err())


def mf_host_spine_err() -> float:
    """After `mf_host_solve`: relative error of the host replay of the spine kernel's front assembly (k_mf_spine) from its own
    plan arrays against the images of the plain recursion (-1: the plan has no spine)."""
    return float(_lib.lib().sqphip_mf_host_spine_err())


class Context:
    """Owns a sqphip_ctx (one NLP structure, `batch` instances)."""

    def __init__(self, n, m, num_linear, jrow, jcol, hrow, hcol, xL, xU, gL, gU,
                 options: _lib.Options | None = None, batch: int = 1):
        self.L = _lib.lib()
        self.n, self.m, self.batch = int(n), int(m), int(batch)
        self.nnzj, self.nnzh = len(jrow), len(hrow)
        self.opts = options or default_options()
        jr, jc, hr, hc = (np.ascontiguousarray(a, dtype=np.int64) for a in (jrow, jcol, hrow, hcol))
        h = C.c_void_p()
        rc = self.L.sqphip_create(C.byref(h), n, m, num_linear, len(jr), _l(jr), _l(jc), len(hr), _l(hr),
                                  _l(hc), _d(_f(xL)), _d(_f(xU)), _d(_f(gL)), _d(_f(gU)),
                                  C.byref(self.opts), batch)
        if rc != 0:
            raise SqpHipError(f"sqphip_create failed with code {rc}")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.sqphip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            msg = self.L.sqphip_last_error(self.h)
            raise SqpHipError(f"libsqphip error {rc}: {msg.decode() if msg else ''}")

    # ---- sub-problem seat
    def qp_solve(self, mode, x_k, delta, mu, df, E, jval, hval):
        p = np.zeros(self.n); lam = np.zeros(self.m)
        mu_u = np.zeros(self.n); mu_l = np.zeros(self.n); slack = np.zeros(2 * self.m)
        st = C.c_int32()
        self._ck(self.L.sqphip_qp_solve(self.h, mode, _d(_f(x_k)), float(delta), float(mu), _d(_f(df)),
                                        _d(_f(E)), _d(_f(jval)),
                                        _d(_f(hval)) if hval is not None and len(hval) else None,
                                        _d(p), _d(lam), _d(mu_u), _d(mu_l), _d(slack), C.byref(st)))
        it, nf = C.c_int32(), C.c_int32()
        self.L.sqphip_qp_stats(self.h, C.byref(it), C.byref(nf))
        rule, err = C.c_int32(), C.c_double()
        self.L.sqphip_qp_termination(self.h, C.byref(rule), C.byref(err))
        return dict(p=p, lam=lam, mult_x_U=mu_u, mult_x_L=mu_l, slack=slack, status=st.value,
                    ipm_iters=it.value, n_factor=nf.value, term_rule=rule.value, scaled_error=err.value)

    def compute_derivative_full(self, df, p, E, mu, mu_vec=None, feasibility_restoration=False, slack=None):
        """compute_derivative(sqp) of sqp.jl:190-213 over merit.jl:13-17 (vector penalty, restoration branch)."""
        out = C.c_double()
        self._ck(self.L.sqphip_compute_derivative_full(self.h, _d(_f(df)), _d(_f(p)), _d(_f(E)), float(mu), _d(_f(mu_vec)),
                                                       int(feasibility_restoration), _d(_f(slack)), C.byref(out)))
        return out.value

    def compute_mu_rule(self, rule, it, rho, x, E, df, p, hval, lam, mu):
        """compute_mu_rule1! / 2! / 3! (sqp_line_search.jl:270-294) with the reductions on the device; returns mu."""
        mu = _f(mu).copy()
        self._ck(self.L.sqphip_compute_mu_rule_dev(self.h, int(rule), int(it), float(rho), _d(_f(x)), _d(_f(E)), _d(_f(df)),
                                                   _d(_f(p)), _d(_f(hval)) if hval is not None and len(hval) else None,
                                                   _d(_f(lam)), _d(mu)))
        return mu

    def acopf_armijo(self, inst, x, p, mu, phi0, D, eta=0.4, tau=0.9, min_alpha=1e-6, feasibility_restoration=False):
        """compute_alpha (sqp_line_search.jl:303-334) on the device: (alpha, is_valid, merit evaluations)."""
        al = C.c_double(); ok = C.c_int32(); ne = C.c_int32()
        self._ck(self.L.sqphip_acopf_armijo(self.h, inst, _d(_f(x)), _d(_f(p)), float(mu), float(phi0), float(D), float(eta),
                                            float(tau), float(min_alpha), int(feasibility_restoration), C.byref(al),
                                            C.byref(ok), C.byref(ne)))
        return al.value, bool(ok.value), ne.value

    def mf_solve_test(self, inst, jval, hval, Dd, sigp, hd, rtype, hsc, dw, rhs):
        """Kernel-level hook of the multifrontal path: (sol_fused, sol_standalone, dinv_by_unknown)."""
        rhs = _f(rhs); a = np.zeros_like(rhs); b = np.zeros_like(rhs); dv = np.zeros_like(rhs)
        rt = np.ascontiguousarray(rtype, dtype=np.int32)
        self._ck(self.L.sqphip_mf_solve_test(self.h, inst, _d(_f(jval)), _d(_f(hval)), _d(_f(Dd)), _d(_f(sigp)),
                                             _d(_f(hd)), _i(rt), float(hsc), float(dw), _d(rhs), _d(a), _d(b), _d(dv)))
        return a, b, dv

    # ---- merit path
    def norm_violations(self, E, x, p=1):
        out = C.c_double()
        pn = {1: 1, 2: 2, math.inf: 0, "inf": 0}[p]
        self._ck(self.L.sqphip_norm_violations(self.h, _d(_f(E)), _d(_f(x)), pn, C.byref(out)))
        return out.value

    def kt_residuals(self, df, lam, mult_x_U, mult_x_L, jval):
        out = C.c_double()
        self._ck(self.L.sqphip_kt_residuals(self.h, _d(_f(df)), _d(_f(lam)), _d(_f(mult_x_U)),
                                            _d(_f(mult_x_L)), _d(_f(jval)), C.byref(out)))
        return out.value

    def norm_complementarity(self, E, lam, p=math.inf):
        out = C.c_double()
        pn = {1: 1, 2: 2, math.inf: 0, "inf": 0}[p]
        self._ck(self.L.sqphip_norm_complementarity(self.h, _d(_f(E)), _d(_f(lam)), pn, C.byref(out)))
        return out.value

    def compute_phi(self, f_trial, E_trial, x_trial, mu, fr):
        out = C.c_double()
        self._ck(self.L.sqphip_compute_phi(self.h, float(f_trial), _d(_f(E_trial)), _d(_f(x_trial)),
                                           float(mu), int(fr), C.byref(out)))
        return out.value

    def compute_qmodel(self, x, p, df, E, jval, hval, mu, with_step):
        out = C.c_double()
        self._ck(self.L.sqphip_compute_qmodel(self.h, _d(_f(x)), _d(_f(p)), _d(_f(df)), _d(_f(E)),
                                              _d(_f(jval)),
                                              _d(_f(hval)) if hval is not None and len(hval) else None,
                                              float(mu), int(with_step), C.byref(out)))
        return out.value

    def compute_derivative(self, df, p, E, mu):
        out = C.c_double()
        self._ck(self.L.sqphip_compute_derivative(self.h, _d(_f(df)), _d(_f(p)), _d(_f(E)), float(mu),
                                                  C.byref(out)))
        return out.value

    def tr_update(self, ared, pred, delta, pnorm, delta_max=1e8):
        acc = C.c_int32(); dn = C.c_double()
        self._ck(self.L.sqphip_tr_update(float(ared), float(pred), float(delta), float(pnorm),
                                         float(delta_max), float(self.opts.tol_direction),
                                         C.byref(acc), C.byref(dn)))
        return bool(acc.value), dn.value

    # ---- ACOPF batch
    def acopf_attach(self, net, lay):
        keep = [np.ascontiguousarray(a, dtype=np.int32) for a in
                (net.f_bus, net.t_bus, net.gen_bus, lay.bal_ptr, lay.bal_colP, lay.bal_colQ)]
        coef = _f(lay.bal_coef)
        form = getattr(lay, "form", "polar")
        if form == "acwr":
            bp = [np.ascontiguousarray(a, dtype=np.int32) for a in (lay.bp_i, lay.bp_j, lay.br_bp)]
            self._ck(self.L.sqphip_acopf_attach_acwr(self.h, net.nb, net.ng, net.nl, *[_i(a) for a in keep], _d(coef),
                                                     int(net.ref_bus), len(bp[0]), *[_i(a) for a in bp], _d(_f(lay.br_sig)),
                                                     _d(_f(lay.bp_tmin)), _d(_f(lay.bp_tmax))))
        else:
            attach = self.L.sqphip_acopf_attach_acr if form == "acr" else self.L.sqphip_acopf_attach
            self._ck(attach(self.h, net.nb, net.ng, net.nl, *[_i(a) for a in keep], _d(coef), int(net.ref_bus)))
        if len(lay.dc_loss1):
            self._ck(self.L.sqphip_acopf_set_dclines(self.h, len(lay.dc_loss1), _d(_f(lay.dc_loss1))))
        if len(lay.sh_bus):
            sb = np.ascontiguousarray(lay.sh_bus, dtype=np.int32)
            self._ck(self.L.sqphip_acopf_set_shunts(self.h, len(sb), _i(sb), _d(_f(lay.sh_gs)), _d(_f(lay.sh_bs))))

    def set_bounds(self, inst, lay):
        """Per-instance variable / row bounds (anything with xL, xU, gL, gU attributes)."""
        self._ck(self.L.sqphip_set_bounds(self.h, inst, _d(_f(lay.xL)), _d(_f(lay.xU)), _d(_f(lay.gL)),
                                          _d(_f(lay.gU))))

    def acopf_set_instance(self, inst, net, lay, x0=None):
        ohm = _f(net.branch_coeffs().ravel())                 # [nl][12], row-major
        self._ck(self.L.sqphip_set_bounds(self.h, inst, _d(_f(lay.xL)), _d(_f(lay.xU)), _d(_f(lay.gL)),
                                          _d(_f(lay.gU))))
        self._ck(self.L.sqphip_acopf_set_instance(self.h, inst, _d(ohm), _d(_f(net.c2)),
                                                  _d(_f(net.c1)), _d(_f(lay.x0 if x0 is None else x0))))

    # ---- the synthetic dense-Hessian NLP (dense_synth.py; csrc/acopf_dev.hpp dense_eval)
    def dense_attach(self, nlp):
        self._ck(self.L.sqphip_dense_attach(self.h, _d(_f(nlp.Q.ravel())), _d(_f(nlp.A.ravel())), float(nlp.kappa)))

    def dense_set_instance(self, inst, nlp, lay, x0=None):
        self._ck(self.L.sqphip_set_bounds(self.h, inst, _d(_f(lay.xL)), _d(_f(lay.xU)), _d(_f(lay.gL)), _d(_f(lay.gU))))
        self._ck(self.L.sqphip_dense_set_instance(self.h, inst, _d(_f(nlp.c)), _d(_f(lay.x0 if x0 is None else x0))))

    def acopf_eval(self, inst, x, sigma=1.0, lam=None):
        f = C.c_double(); grad = np.zeros(self.n); g = np.zeros(self.m)
        jv = np.zeros(self.nnzj); hv = np.zeros(self.nnzh) if lam is not None else None
        self._ck(self.L.sqphip_acopf_eval(self.h, inst, _d(_f(x)), float(sigma), _d(_f(lam)), C.byref(f),
                                          _d(grad), _d(g), _d(jv), _d(hv)))
        return dict(f=f.value, grad=grad, g=g, jval=jv, hval=hv)

    def sqp_reset(self):
        self._ck(self.L.sqphip_sqp_reset(self.h))

    def sqp_run(self, max_outer=0):
        self._ck(self.L.sqphip_sqp_run(self.h, int(max_outer)))

    def sqp_get(self, inst):
        x = np.zeros(self.n); g = np.zeros(self.m); mg = np.zeros(self.m)
        ml = np.zeros(self.n); mu = np.zeros(self.n)
        obj = C.c_double(); st = C.c_int32(); it = C.c_int32()
        self._ck(self.L.sqphip_sqp_get(self.h, inst, _d(x), _d(g), _d(mg), _d(ml), _d(mu), C.byref(obj),
                                       C.byref(st), C.byref(it)))
        return dict(x=x, g=g, mult_g=mg, mult_x_L=ml, mult_x_U=mu, obj_val=obj.value, status=st.value,
                    iter=it.value)

    def sqp_status(self):
        ret = np.zeros(self.batch, dtype=np.int32); it = np.zeros(self.batch, dtype=np.int32)
        done = np.zeros(self.batch, dtype=np.int32)
        self._ck(self.L.sqphip_sqp_status(self.h, _i(ret), _i(it), _i(done)))
        return ret, it, done

    def sqp_trace(self, inst, cap=4096):
        rows = np.zeros((cap, 12)); n = C.c_int32()
        self._ck(self.L.sqphip_sqp_trace(self.h, inst, _d(rows), cap, C.byref(n)))
        names = ("iter", "accepted", "fr", "sub_status", "ipm_iters", "f", "phi", "mu", "delta", "pnorm",
                 "prim_infeas", "dual_infeas")
        out = []
        for k in range(min(n.value, cap)):
            r = dict(zip(names, rows[k]))
            for key in names[:5]:
                r[key] = int(r[key])
            out.append(r)
        return out

    # ---- multi-GPU status gather (RCCL inside the library)
    @staticmethod
    def comm_available() -> bool:
        """librccl loads in this process (ask on every rank and agree before the collective comm_init)"""
        return bool(_lib.lib().sqphip_comm_available())

    @staticmethod
    def comm_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        rc = _lib.lib().sqphip_comm_unique_id(C.cast(buf, C.c_void_p))
        if rc != 0:
            raise SqpHipError(f"sqphip_comm_unique_id failed ({rc})")
        return buf.raw

    def comm_init(self, unique_id: bytes, world: int, rank: int):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        self._ck(self.L.sqphip_comm_init(self.h, C.cast(buf, C.c_void_p), int(world), int(rank)))

    def gather_status(self, total: int):
        ret = np.zeros(total, dtype=np.int32); it = np.zeros(total, dtype=np.int32); done = np.zeros(total, dtype=np.int32)
        self._ck(self.L.sqphip_gather_status(self.h, int(total), _i(ret), _i(it), _i(done)))
        return ret, it, done

    def comm_destroy(self):
        self._ck(self.L.sqphip_comm_destroy(self.h))

    def counters(self):
        c = _lib.Counters()
        self._ck(self.L.sqphip_get_counters(self.h, C.byref(c)))
        return {k: getattr(c, k) for k, _ in _lib.Counters._fields_}

    def mode_counters(self):
        """Work of the batched run since sqp_reset by sub-problem mode: {mode: (sub-problems, IPM iterations,
        factorisations)} for QP / FR / SOC / LP."""
        out = (C.c_int64 * 12)()
        self._ck(self.L.sqphip_get_mode_counters(self.h, out))
        return {name: (out[3 * k], out[3 * k + 1], out[3 * k + 2]) for k, name in enumerate(("QP", "FR", "SOC", "LP"))}

    def sqp_work(self):
        """Per instance: (sub-problems, IPM iterations, factorisations) since sqp_reset, three int64 arrays."""
        out = [np.zeros(self.batch, dtype=np.int64) for _ in range(3)]
        self._ck(self.L.sqphip_sqp_work(self.h, *[a.ctypes.data_as(C.POINTER(C.c_int64)) for a in out]))
        return tuple(out)

    def sqp_qp_log(self, inst):
        """The last (up to 64) sub-problems of an instance: list of (mode, MOI status, IPM iterations, factorisations)."""
        rows = np.zeros((64, 4), dtype=np.int32); n = C.c_int32()
        self._ck(self.L.sqphip_sqp_qp_log(self.h, inst, _i(rows), 64, C.byref(n)))
        return [tuple(int(v) for v in rows[k]) for k in range(n.value)]

    def sqp_qp_log_term(self, inst):
        """For the rows of sqp_qp_log: (final scaled optimality error, rule that ended the interior-point run: 0 tolerance,
        1 / 2 / 3 acceptable-termination rules, -1 not converged)."""
        err = np.zeros(64); rule = np.zeros(64, dtype=np.int32); n = C.c_int32()
        self._ck(self.L.sqphip_sqp_qp_log_term(self.h, inst, _d(err), _i(rule), 64, C.byref(n)))
        return [(float(err[k]), int(rule[k])) for k in range(n.value)]

    def termination_counters(self):
        """Sub-problems of the batched run since sqp_reset ended by (tolerance, rule 1, rule 2, rule 3)."""
        out = (C.c_int64 * 4)()
        self._ck(self.L.sqphip_get_termination_counters(self.h, out))
        return tuple(int(v) for v in out)

    def sqp_last_request(self, inst):
        """The sub-problem request an instance of the batched run worked on last (arguments of QpHip / sqphip_qp_solve)."""
        mode = C.c_int32(); delta = C.c_double(); mu = C.c_double()
        xk = np.zeros(self.n); c = np.zeros(self.n); b = np.zeros(self.m); jc = np.zeros(self.nnzj); hc = np.zeros(self.nnzh)
        self._ck(self.L.sqphip_sqp_last_request(self.h, inst, C.byref(mode), C.byref(delta), C.byref(mu), _d(xk), _d(c),
                                                _d(b), _d(jc), _d(hc)))
        return dict(mode=mode.value, delta=delta.value, mu_pen=mu.value, x_k=xk, c=c, b=b, jac_coo=jc, hess_coo=hc)

    # ---- scenario queue (more scenarios than slots)
    def stream_begin(self, n_scenarios):
        self._ck(self.L.sqphip_sqp_stream_begin(self.h, int(n_scenarios)))

    def stream_set(self, scen, net, lay, x0=None):
        self._ck(self.L.sqphip_sqp_stream_set(self.h, int(scen), _d(_f(lay.xL)), _d(_f(lay.xU)), _d(_f(lay.gL)), _d(_f(lay.gU)),
                                              _d(_f(net.branch_coeffs().ravel())), _d(_f(net.c2)), _d(_f(net.c1)),
                                              _d(_f(lay.x0 if x0 is None else x0))))

    def stream_run(self):
        self._ck(self.L.sqphip_sqp_stream_run(self.h))

    # ... shared between ranks: this rank's part of the queue as an explicit id list
    def stream_assign(self, ids):
        a = np.ascontiguousarray(ids, dtype=np.int32)
        self._ck(self.L.sqphip_sqp_stream_assign(self.h, len(a), _i(a)))

    def stream_append(self, ids):
        a = np.ascontiguousarray(ids, dtype=np.int32)
        self._ck(self.L.sqphip_sqp_stream_append(self.h, len(a), _i(a)))

    def stream_release(self, n):
        """take up to n unstarted ids off the tail of this rank's queue"""
        out = np.zeros(max(1, int(n)), dtype=np.int32); k = C.c_int32()
        self._ck(self.L.sqphip_sqp_stream_release(self.h, int(n), _i(out), C.byref(k)))
        return out[:k.value].copy()

    def stream_run_some(self, max_outer):
        """every slot performs up to max_outer more outer iterations; returns (unstarted ids of this rank, slots still running)"""
        u, a = C.c_int32(), C.c_int32()
        self._ck(self.L.sqphip_sqp_stream_run_some(self.h, int(max_outer), C.byref(u), C.byref(a)))
        return u.value, a.value

    def stream_get(self, scen):
        x = np.zeros(self.n); obj = C.c_double(); st = C.c_int32(); it = C.c_int32()
        self._ck(self.L.sqphip_sqp_stream_get(self.h, int(scen), _d(x), C.byref(obj), C.byref(st), C.byref(it)))
        return dict(x=x, obj_val=obj.value, status=st.value, iter=it.value)

    def reset_counters(self):
        self._ck(self.L.sqphip_reset_counters(self.h))

    KERNEL_CLASSES = ("values", "fronts_low", "fronts_top", "solve_top", "solve_levels", "post", "transitions")

    def kernel_times(self):
        """{class: (seconds of kernel time, launch groups timed)} since reset_counters (set_timing(2) collects them)."""
        sec = np.zeros(7); grp = np.zeros(7, dtype=np.int64)
        self._ck(self.L.sqphip_get_kernel_times(self.h, _d(sec), grp.ctypes.data_as(C.POINTER(C.c_int64)), 7))
        return {k: (float(sec[i]), int(grp[i])) for i, k in enumerate(self.KERNEL_CLASSES)}

    def set_timing(self, on: bool):
        self._ck(self.L.sqphip_set_timing(self.h, int(on)))


# ------------------------------------------------------------------------------------------------
@dataclasses.dataclass
class QpData:
    """subproblem.jl:12-23.  Q / A are carried as COO values in the structure order of the Model
    (what eval_h / eval_jac_g fill); the library merges duplicates and mirrors the Hessian."""
    Q: np.ndarray | None
    c: np.ndarray
    A: np.ndarray
    b: np.ndarray
    c_lb: np.ndarray
    c_ub: np.ndarray
    v_lb: np.ndarray
    v_ub: np.ndarray
    num_linear_constraints: int


class QpHip:
    """`AbstractSubOptimizer` backed by libsqphip; method names and 6-tuple returns follow QpJuMP."""

    def __init__(self, ctx: Context, data: QpData | None = None):
        self.ctx = ctx
        self.data = data

    def create_model(self, delta):          # subproblem_JuMP.jl:36-125: nothing to build, the ctx is the model
        return None

    def _solve(self, mode, x_k, delta, mu=1.0):
        dta = self.data
        r = self.ctx.qp_solve(mode, x_k, delta, mu, dta.c, dta.b, dta.A, dta.Q)
        return r["p"], r["lam"], r["mult_x_U"], r["mult_x_L"], r["slack"], r["status"]

    def sub_optimize(self, x_k, delta):
        return self._solve(MODE_QP, x_k, delta)

    def sub_optimize_FR(self, x_k, delta):
        return self._solve(MODE_FR, x_k, delta)

    def sub_optimize_L1QP(self, x_k, delta, mu):
        return self._solve(MODE_L1QP, x_k, delta, mu)

    def sub_optimize_infeas(self, x_k, delta):
        p, _, _, _, slack, st = self._solve(MODE_INFEAS, x_k, delta)
        return p, (float(slack.sum()) if st in _OK else math.inf)

    def sub_optimize_lp(self, x_k):
        p, lam, mu_u, mu_l, _, st = self._solve(MODE_LP, x_k, math.inf)
        return p, lam, mu_u, mu_l, st
