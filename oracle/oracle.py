"""ctypes binding of the CPU oracle (oracle/build/liboracle.so).

TEST INFRASTRUCTURE ONLY: import this from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never from the product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "liboracle.so")
_SRCS = ["common.c", "qp_ipm.c", "sqp_tr.c", "problems.c", "sparse_ldlt.c", "sqp_oracle.h"]

MODE_QP, MODE_FR, MODE_SOC, MODE_LP, MODE_L1QP, MODE_INFEAS = range(6)
MOI_LOCALLY_SOLVED, MOI_LOCALLY_INFEASIBLE, MOI_ITERATION_LIMIT = 4, 5, 11


class Options(C.Structure):
    _fields_ = [(k, C.c_double) for k in
                ("tol_direction", "tol_residual", "tol_infeas", "init_mu", "max_mu", "tr_size",
                 "rho", "eta", "tau", "min_alpha")] + \
               [("max_iter", C.c_int), ("use_soc", C.c_int), ("literal_quirks", C.c_int),
                ("ipm_tol", C.c_double),
                ("ipm_max_iter", C.c_int), ("ipm_phase1", C.c_int), ("num_threads", C.c_int),
                ("ipm_corrector", C.c_int), ("kkt_condense", C.c_int), ("kkt_tile_order", C.c_int),
                ("kkt_mode", C.c_int), ("ipm_warm_start", C.c_int)]


class TraceRow(C.Structure):
    _fields_ = [("iter", C.c_int), ("accepted", C.c_int), ("fr", C.c_int),
                ("sub_status", C.c_int), ("ipm_iters", C.c_int), ("n_factor", C.c_int)] + \
               [(k, C.c_double) for k in
                ("f", "phi", "mu", "delta", "pnorm", "prim_infeas", "dual_infeas")]


class Result(C.Structure):
    _fields_ = [("status", C.c_int), ("iter", C.c_int), ("obj_val", C.c_double),
                ("n_qp", C.c_int), ("n_ipm_iter", C.c_int), ("n_factor", C.c_int),
                ("qp_seconds", C.c_double), ("trace_len", C.c_int)]


class Nlp(C.Structure):
    _fields_ = [("n", C.c_int64), ("m", C.c_int64), ("num_linear", C.c_int64),
                ("nnzj", C.c_int64), ("nnzh", C.c_int64),
                ("jrow", C.POINTER(C.c_int64)), ("jcol", C.POINTER(C.c_int64)),
                ("hrow", C.POINTER(C.c_int64)), ("hcol", C.POINTER(C.c_int64)),
                ("xL", C.POINTER(C.c_double)), ("xU", C.POINTER(C.c_double)),
                ("gL", C.POINTER(C.c_double)), ("gU", C.POINTER(C.c_double)),
                ("eval_f", C.c_void_p), ("eval_grad_f", C.c_void_p), ("eval_g", C.c_void_p),
                ("eval_jac_g", C.c_void_p), ("eval_h", C.c_void_p), ("ud", C.c_void_p)]


_lib = None


def build(force: bool = False) -> str:
    """Compile the oracle if the .so is missing or older than its sources."""
    stale = force or not os.path.exists(_SO)
    if not stale:
        t = os.path.getmtime(_SO)
        stale = any(os.path.getmtime(os.path.join(_HERE, s)) > t for s in _SRCS)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "all"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        dp, ip, lp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        L.ora_default_options.argtypes = [C.POINTER(Options)]
        L.ora_norm_violations.restype = C.c_double
        L.ora_norm_violations.argtypes = [C.c_int64, C.c_int64, dp, dp, dp, dp, dp, dp, C.c_int]
        L.ora_kt_residuals.restype = C.c_double
        L.ora_kt_residuals.argtypes = [C.c_int64, C.c_int64, dp, dp, dp, dp, lp, lp, dp]
        L.ora_norm_complementarity.restype = C.c_double
        L.ora_norm_complementarity.argtypes = [C.c_int64, dp, dp, dp, dp, C.c_int]
        L.ora_compute_derivative.restype = C.c_double
        L.ora_compute_derivative.argtypes = [C.c_double, C.c_double, C.c_int64, dp]
        L.ora_compute_derivative_full.restype = C.c_double
        L.ora_compute_derivative_full.argtypes = [C.c_int64, C.c_int64, dp, dp, dp, dp, dp, C.c_double, dp, C.c_int, dp, C.c_int64]
        L.ora_compute_mu_rule.argtypes = [C.c_int, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64, dp, dp]
        L.ora_armijo_alpha.restype = C.c_double
        L.ora_isapprox.restype = C.c_int
        L.ora_isapprox.argtypes = [C.c_double, C.c_double]
        L.ora_qp_create.restype = C.c_void_p
        L.ora_qp_create.argtypes = [C.c_int64, C.c_int64, C.c_int64, lp, lp, lp, lp,
                                    dp, dp, dp, dp, C.POINTER(Options)]
        L.ora_qp_destroy.argtypes = [C.c_void_p]
        L.ora_qp_solve.restype = C.c_int
        L.ora_qp_solve.argtypes = [C.c_void_p, C.c_int, dp, C.c_double, C.c_double,
                                   dp, dp, dp, dp, dp, dp, dp, dp, dp]
        L.ora_qp_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), dp]
        L.ora_ldlt_factor.argtypes = [C.c_int64, dp, C.c_int64, dp, C.c_int64, lp, lp, C.c_int]
        L.ora_ldlt_solve.argtypes = [C.c_int64, dp, C.c_int64, dp, dp]
        L.ora_sqp_tr_solve.argtypes = [C.POINTER(Nlp), C.POINTER(Options), dp, dp, dp, dp, dp,
                                       C.POINTER(Result), C.POINTER(TraceRow), C.c_int]
        for nm in ("ora_problem_toy", "ora_problem_readme1", "ora_problem_hs071"):
            getattr(L, nm).restype = C.c_void_p
        for ctor in (L.ora_problem_acopf, L.ora_problem_acopf_acr):
            ctor.restype = C.c_void_p
            ctor.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip, dp, ip, dp, dp,
                             ip, ip, ip, dp, C.c_int64, lp, lp, C.c_int64, lp, lp,
                             dp, dp, dp, dp, C.c_int, ip, dp, dp, C.c_int, dp]
        L.ora_problem_acopf_acwr.restype = C.c_void_p
        L.ora_problem_acopf_acwr.argtypes = L.ora_problem_acopf.argtypes + [C.c_int, ip, ip, ip, dp, dp, dp]
        L.ora_problem_dense.restype = C.c_void_p
        L.ora_problem_dense.argtypes = [C.c_int64, C.c_int64, dp, dp, dp, C.c_double, C.c_int64, lp, lp, C.c_int64, lp, lp, dp, dp, dp, dp, dp]
        L.ora_problem_nlp.restype = C.POINTER(Nlp)
        L.ora_problem_nlp.argtypes = [C.c_void_p]
        L.ora_problem_x0.restype = dp
        L.ora_problem_x0.argtypes = [C.c_void_p]
        L.ora_problem_destroy.argtypes = [C.c_void_p]
        L.ora_problem_drop_hessian.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _l(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def set_kkt_order(pos=None):
    """Order of the condensed Newton matrix for every solver created afterwards: `pos` as returned by the product
    library's host-only `sqphip_kkt_order` (padding is squeezed out here); None restores the natural order."""
    L = lib()
    L.ora_set_kkt_order.argtypes = [C.POINTER(C.c_int32), C.c_int64]
    if pos is None:
        L.ora_set_kkt_order(None, 0); return
    rank = np.empty(len(pos), dtype=np.int32); rank[np.argsort(pos, kind="stable")] = np.arange(len(pos), dtype=np.int32)
    L.ora_set_kkt_order(rank.ctypes.data_as(C.POINTER(C.c_int32)), len(rank))


def _apply_order(opts, n, m, jrow1, jcol1, hrow1, hcol1, gL, gU):
    """opts.kkt_tile_order: factorise in the product library's order.  The permutation comes from the library's
    host-only, GPU-free `sqphip_kkt_order` (pure integer work on the sparsity pattern); every permutation gives the
    same mathematics (tests/test_oracle_kat.py compares it with the natural order), following the product's one only
    keeps the two implementations on one rounding trajectory."""
    if not (opts.kkt_condense and opts.kkt_tile_order):
        return                  # the oracle's own order (natural for the dense LDL^T, its minimum degree for the sparse one)
    import sqpsolver_jl_amd as _pkg
    pos, _, _ = _pkg.kkt_order(int(n), int(m), jrow1, jcol1, hrow1, hcol1, f64(gL), f64(gU))
    set_kkt_order(pos)


def default_options(**kw) -> Options:
    o = Options()
    lib().ora_default_options(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class Problem:
    """Owns an ora_problem handle."""

    def __init__(self, handle, x0=None):
        self.h = handle
        self.nlp = lib().ora_problem_nlp(handle)
        self.n = int(self.nlp.contents.n)
        self.m = int(self.nlp.contents.m)
        px = lib().ora_problem_x0(handle)
        self.x0 = np.array([px[i] for i in range(self.n)]) if x0 is None else f64(x0).copy()

    def __del__(self):
        try:
            lib().ora_problem_destroy(self.h)
        except Exception:
            pass

    # raw callback access (for derivative checks)
    def _fn(self, name, proto):
        return proto(getattr(self.nlp.contents, name))

    def eval_f(self, x):
        fn = self._fn("eval_f", C.CFUNCTYPE(C.c_double, C.c_void_p, C.POINTER(C.c_double)))
        return fn(self.nlp.contents.ud, _d(f64(x)))

    def eval_grad_f(self, x):
        out = np.zeros(self.n)
        fn = self._fn("eval_grad_f", C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double),
                                                 C.POINTER(C.c_double)))
        fn(self.nlp.contents.ud, _d(f64(x)), _d(out))
        return out

    def eval_g(self, x):
        out = np.zeros(self.m)
        fn = self._fn("eval_g", C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double),
                                            C.POINTER(C.c_double)))
        fn(self.nlp.contents.ud, _d(f64(x)), _d(out))
        return out

    def eval_jac_g(self, x):
        out = np.zeros(int(self.nlp.contents.nnzj))
        fn = self._fn("eval_jac_g", C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double),
                                                C.POINTER(C.c_double)))
        fn(self.nlp.contents.ud, _d(f64(x)), _d(out))
        return out

    def eval_h(self, x, sigma, lam):
        out = np.zeros(int(self.nlp.contents.nnzh))
        fn = self._fn("eval_h", C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.c_double,
                                            C.POINTER(C.c_double), C.POINTER(C.c_double)))
        fn(self.nlp.contents.ud, _d(f64(x)), float(sigma), _d(f64(lam)), _d(out))
        return out

    def structure(self):
        c = self.nlp.contents
        nj, nh = int(c.nnzj), int(c.nnzh)
        g = lambda p, k: np.array([p[i] for i in range(k)])
        return dict(n=self.n, m=self.m, num_linear=int(c.num_linear),
                    jrow=g(c.jrow, nj), jcol=g(c.jcol, nj), hrow=g(c.hrow, nh), hcol=g(c.hcol, nh),
                    xL=g(c.xL, self.n), xU=g(c.xU, self.n), gL=g(c.gL, self.m), gU=g(c.gU, self.m))


def drop_hessian(prob: Problem) -> Problem:
    """The same problem without second derivatives (the reference's `eval_h === nothing` path, sqp.jl:92): every
    sub-problem gets a linear objective (subproblem_JuMP.jl:137-140)."""
    lib().ora_problem_drop_hessian(prob.h)
    return prob


def problem_toy():
    return Problem(lib().ora_problem_toy())


def problem_readme1():
    return Problem(lib().ora_problem_readme1())


def problem_hs071():
    return Problem(lib().ora_problem_hs071())


def problem_acopf(net, lay):
    ohm = f64(net.branch_coeffs().ravel())                    # [nl][12], row-major
    keep = [np.ascontiguousarray(a) for a in
            (net.f_bus.astype(np.int32), net.t_bus.astype(np.int32), net.gen_bus.astype(np.int32),
             lay.bal_ptr.astype(np.int32), lay.bal_colP.astype(np.int32),
             lay.bal_colQ.astype(np.int32))]
    fb, tb, gb, bp, bcp, bcq = keep
    args = [f64(a) for a in (net.c2, net.c1, lay.bal_coef, lay.xL, lay.xU, lay.gL, lay.gU)]
    c2, c1, coef, xL, xU, gL, gU = args
    jr, jc, hr, hc = (np.ascontiguousarray(a, dtype=np.int64)
                      for a in (lay.jrow, lay.jcol, lay.hrow, lay.hcol))
    form = getattr(lay, "form", "polar")
    common = (net.nb, net.ng, net.nl, _i(fb), _i(tb), _d(ohm),
              _i(gb), _d(c2), _d(c1), _i(bp), _i(bcp), _i(bcq), _d(coef),
              len(jr), _l(jr), _l(jc), len(hr), _l(hr), _l(hc),
              _d(xL), _d(xU), _d(gL), _d(gU), len(lay.sh_bus),
              _i(np.ascontiguousarray(lay.sh_bus, dtype=np.int32)), _d(f64(lay.sh_gs)), _d(f64(lay.sh_bs)),
              len(lay.dc_loss1), _d(f64(lay.dc_loss1)))
    if form == "acwr":
        bpi, bpj, brb = (np.ascontiguousarray(a, dtype=np.int32) for a in (lay.bp_i, lay.bp_j, lay.br_bp))
        h = lib().ora_problem_acopf_acwr(*common, len(bpi), _i(bpi), _i(bpj), _i(brb), _d(f64(lay.br_sig)),
                                         _d(f64(lay.bp_tmin)), _d(f64(lay.bp_tmax)))
    else:
        h = (lib().ora_problem_acopf_acr if form == "acr" else lib().ora_problem_acopf)(*common)
    return Problem(h, x0=lay.x0)


def problem_dense(nlp, lay):
    """The synthetic dense-Hessian NLP of sqpsolver.jl_amd/dense_synth.py (CPU twin of the device evaluator dense_eval)."""
    jr, jc, hr, hc = (np.ascontiguousarray(a, dtype=np.int64) for a in (lay.jrow, lay.jcol, lay.hrow, lay.hcol))
    args = [f64(a) for a in (nlp.Q.ravel(), nlp.A.ravel(), nlp.c, lay.xL, lay.xU, lay.gL, lay.gU, lay.x0)]
    Q, A, c, xL, xU, gL, gU, x0 = args
    h = lib().ora_problem_dense(nlp.n, nlp.m, _d(Q), _d(A), _d(c), float(nlp.kappa), len(jr), _l(jr), _l(jc), len(hr), _l(hr), _l(hc),
                                _d(xL), _d(xU), _d(gL), _d(gU), _d(x0))
    if not h:
        raise ValueError("ora_problem_dense: structure does not match the dense layout")
    return Problem(h, x0=lay.x0)


def sqp_solve(prob: Problem, opts: Options | None = None, x0=None, trace_cap: int = 4096):
    """Run the restated SqpTR.run! (sqp_trust_region.jl:98-223) and return everything."""
    opts = opts or default_options()
    S = prob.structure()
    _apply_order(opts, prob.n, prob.m, S["jrow"], S["jcol"], S["hrow"], S["hcol"], S["gL"], S["gU"])
    x = f64(prob.x0 if x0 is None else x0).copy()
    g = np.zeros(prob.m)
    mg = np.zeros(prob.m)
    mxl = np.zeros(prob.n)
    mxu = np.zeros(prob.n)
    res = Result()
    tr = (TraceRow * trace_cap)()
    lib().ora_sqp_tr_solve(prob.nlp, C.byref(opts), _d(x), _d(g), _d(mg), _d(mxl), _d(mxu),
                           C.byref(res), tr, trace_cap)
    names = [f[0] for f in TraceRow._fields_]
    trace = [{k: getattr(tr[i], k) for k in names} for i in range(min(res.trace_len, trace_cap))]
    return dict(x=x, g=g, mult_g=mg, mult_x_L=mxl, mult_x_U=mxu, status=res.status,
                iter=res.iter, obj_val=res.obj_val, n_qp=res.n_qp, n_ipm_iter=res.n_ipm_iter,
                n_factor=res.n_factor, qp_seconds=res.qp_seconds, trace=trace)


def coo_to_csc(n_cols, rows1, cols1, sym=False):
    """Pattern + COO->slot maps the way Julia's sparse(I,J,V) merges duplicates
    (sqp_trust_region.jl:47-48,56-57); sym mirrors off-diagonals (sqp.jl:96-101)."""
    r = np.asarray(rows1, dtype=np.int64) - 1
    c = np.asarray(cols1, dtype=np.int64) - 1
    k = np.arange(len(r))
    if sym:
        off = r != c
        rr = np.concatenate([r, c[off]])
        cc = np.concatenate([c, r[off]])
        kk = np.concatenate([k, k[off]])
        tt = np.concatenate([np.zeros(len(r), int), np.ones(off.sum(), int)])
    else:
        rr, cc, kk, tt = r, c, k, np.zeros(len(r), int)
    order = np.lexsort((rr, cc))
    rr, cc, kk, tt = rr[order], cc[order], kk[order], tt[order]
    new = np.ones(len(rr), bool)
    new[1:] = (rr[1:] != rr[:-1]) | (cc[1:] != cc[:-1])
    slot_of = np.cumsum(new) - 1
    rowval = rr[new]
    colptr = np.zeros(n_cols + 1, dtype=np.int64)
    np.add.at(colptr, cc[new] + 1, 1)
    colptr = np.cumsum(colptr)
    slot = np.full(len(r), -1, dtype=np.int64)
    slot_t = np.full(len(r), -1, dtype=np.int64)
    slot[kk[tt == 0]] = slot_of[tt == 0]
    slot_t[kk[tt == 1]] = slot_of[tt == 1]
    return colptr, rowval.astype(np.int64), slot, slot_t


class QpSolver:
    """The sub-problem seat: ora_qp_* (subproblem_JuMP.jl QpJuMP stand-in)."""

    def __init__(self, n, m, num_linear, jcolptr, jrowval, hcolptr, hrowval, xL, xU, gL, gU,
                 opts: Options | None = None):
        self.n, self.m = n, m
        self.opts = opts or default_options()
        self._keep = [np.ascontiguousarray(a, dtype=np.int64)
                      for a in (jcolptr, jrowval, hcolptr, hrowval)]
        jc, jr, hc, hr = self._keep
        csc_cols = lambda ptr: np.repeat(np.arange(n, dtype=np.int64), np.diff(ptr))
        _apply_order(self.opts, n, m, jr + 1, csc_cols(jc) + 1, hr + 1, csc_cols(hc) + 1, gL, gU)
        self.h = lib().ora_qp_create(n, m, num_linear, _l(jc), _l(jr), _l(hc), _l(hr),
                                     _d(f64(xL)), _d(f64(xU)), _d(f64(gL)), _d(f64(gU)),
                                     C.byref(self.opts))

    def __del__(self):
        try:
            lib().ora_qp_destroy(self.h)
        except Exception:
            pass

    def solve(self, mode, x_k, delta, mu, c, b, jval, hval, want_slack=False):
        p = np.zeros(self.n); lam = np.zeros(self.m)
        mu_u = np.zeros(self.n); mu_l = np.zeros(self.n)
        slack = np.zeros(2 * self.m)
        nul = C.POINTER(C.c_double)()
        st = lib().ora_qp_solve(self.h, mode, _d(f64(x_k)), float(delta), float(mu),
                                _d(f64(c)) if c is not None else nul,
                                _d(f64(b)) if b is not None else nul, _d(f64(jval)),
                                _d(f64(hval)) if hval is not None and len(hval) else nul,
                                _d(p), _d(lam), _d(mu_u), _d(mu_l), _d(slack))
        it, nf = C.c_int(), C.c_int()
        el = C.c_double()
        lib().ora_qp_stats(self.h, C.byref(it), C.byref(nf), C.byref(el))
        rule, err = C.c_int(), C.c_double()
        lib().ora_qp_termination(C.c_void_p(self.h), C.byref(rule), C.byref(err))     # how the interior-point run ended (qp_ipm.c)
        out = dict(status=st, p=p, lam=lam, mult_x_U=mu_u, mult_x_L=mu_l,
                   ipm_iters=it.value, n_factor=nf.value, elastic=el.value, term_rule=rule.value, scaled_error=err.value)
        if want_slack:
            out["slack"] = slack
        return out


def norm_violations(E, gL, gU, x, xL, xU, p=1):
    pn = {1: 1, 2: 2, np.inf: 0, "inf": 0}[p]
    return lib().ora_norm_violations(len(E), len(x), _d(f64(E)), _d(f64(gL)), _d(f64(gU)),
                                     _d(f64(x)), _d(f64(xL)), _d(f64(xU)), pn)


def kt_residuals(df, lam, mxu, mxl, colptr, rowval, nzval, m):
    cp = np.ascontiguousarray(colptr, dtype=np.int64)
    rv = np.ascontiguousarray(rowval, dtype=np.int64)
    return lib().ora_kt_residuals(m, len(df), _d(f64(df)), _d(f64(lam)), _d(f64(mxu)),
                                  _d(f64(mxl)), _l(cp), _l(rv), _d(f64(nzval)))


def ldlt_factor(A, n1, nthreads=1):
    """A: (N,N) symmetric; returns (L-with-unit-diagonal lower incl. garbage upper, dinv, npos1, nneg2)."""
    N = A.shape[0]
    ld = N
    a = np.asfortranarray(A, dtype=np.float64).copy(order="F")
    dinv = np.zeros(N)
    npos, nneg = C.c_int64(), C.c_int64()
    lib().ora_ldlt_factor(N, _d(a), ld, _d(dinv), n1, C.byref(npos), C.byref(nneg), nthreads)
    return a, dinv, npos.value, nneg.value


def ldlt_solve(a, dinv, rhs):
    N = a.shape[0]
    x = f64(rhs).copy()
    lib().ora_ldlt_solve(N, _d(a), N, _d(dinv), _d(x))
    return x
