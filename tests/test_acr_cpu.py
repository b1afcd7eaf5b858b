"""Rectangular-voltage (ACR) formulation of the ACOPF evaluator (SURVEY.md section 8f-4;
/root/reference/examples/acopf/opf.jl:46,51 -- the formulation run_sqp_opf instantiates): layout, the oracle's
callbacks (test infrastructure) and the library's host-side symbolic analysis on the ACR structure.  CPU only."""
import dataclasses

import numpy as np
import pytest

import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, acr_layout, acwr_layout, contingency, CASES
from oracle import oracle as O


def _net(case="case14", shunts=False):
    nb, ng, nl, seed = CASES[case]
    net = contingency(acopf_synth(nb, ng, nl, seed), 5, seed)
    if shunts:
        rng = np.random.default_rng(seed)
        net = dataclasses.replace(net, gs=np.where(rng.random(nb) < 0.3, rng.uniform(0, 0.03, nb), 0.0),
                                  bs=np.where(rng.random(nb) < 0.4, rng.uniform(-0.05, 0.19, nb), 0.0),
                                  tap=np.where(rng.random(nl) < 0.3, rng.uniform(0.93, 1.07, nl), 1.0),
                                  shift=np.where(rng.random(nl) < 0.1, rng.uniform(-0.08, 0.08, nl), 0.0))
    return net


def _dense(lay, vals, kind):
    if kind == "J":
        A = np.zeros((lay.m, lay.n)); np.add.at(A, (lay.jrow - 1, lay.jcol - 1), vals)
        return A
    A = np.zeros((lay.n, lay.n)); np.add.at(A, (lay.hrow - 1, lay.hcol - 1), vals)
    return A + A.T - np.diag(np.diag(A))


@pytest.mark.parametrize("shunts", [False, True])
def test_acr_layout_counts_and_polar_equivalence(shunts):
    """Structure counts as sqphip_acopf_attach_acr checks them; at corresponding points (vr + j vi = vm e^{j va}) the
    Ohm and balance rows of the two formulations have the same values and vr^2 + vi^2 = vm^2."""
    net = _net(shunts=shunts)
    nb, ng, nl = net.nb, net.ng, net.nl
    lp, lr = acopf_layout(net), acr_layout(net)
    nsh = len(lr.sh_bus)
    assert (nsh > 0) == shunts
    assert lr.form == "acr" and lr.n == lp.n and lr.m == 1 + 4 * nb + 6 * nl
    assert len(lr.jrow) == 1 + 2 * (2 * nl + ng) + 4 * nb + 24 * nl + 4 * nsh
    assert len(lr.hrow) == ng + 28 * nl + 4 * nb + 2 * nsh and (lr.hrow >= lr.hcol).all()
    assert lr.num_linear == (1 if shunts else 1 + 2 * nb)
    Pp, Pr = O.problem_acopf(net, lp), O.problem_acopf(net, lr)
    rng = np.random.default_rng(0)
    xp = lp.x0 + 0.05 * rng.standard_normal(lp.n)
    va, vm = xp[:nb], xp[nb:2 * nb]
    xr = xp.copy(); xr[:nb] = vm * np.sin(va); xr[nb:2 * nb] = vm * np.cos(va)
    gp, gr = Pp.eval_g(xp), Pr.eval_g(xr)
    O0p, O0r = 2 * nl + 1 + 2 * nb + 2 * nl, 1 + 4 * nb + 2 * nl
    assert np.abs(gp[O0p:O0p + 4 * nl] - gr[O0r:O0r + 4 * nl]).max() < 1e-13
    assert np.abs(gp[2 * nl + 1:2 * nl + 1 + 2 * nb] - gr[1:1 + 2 * nb]).max() < 1e-13
    assert np.abs(gr[1 + 2 * nb:1 + 4 * nb:2] - vm ** 2).max() < 1e-14
    assert Pp.eval_f(xp) == Pr.eval_f(xr)


@pytest.mark.parametrize("shunts", [False, True])
def test_acr_derivatives_against_finite_differences(shunts):
    net = _net(shunts=shunts)
    lay = acr_layout(net)
    P = O.problem_acopf(net, lay)
    rng = np.random.default_rng(1)
    x = lay.x0 + 0.05 * rng.standard_normal(lay.n); lam = rng.standard_normal(lay.m); sig = 0.7
    h = 1e-6
    J = _dense(lay, P.eval_jac_g(x), "J")
    Jfd = np.zeros_like(J)
    for j in range(lay.n):
        e = np.zeros(lay.n); e[j] = h
        Jfd[:, j] = (P.eval_g(x + e) - P.eval_g(x - e)) / (2 * h)
    assert np.abs(J - Jfd).max() < 1e-7
    H = _dense(lay, P.eval_h(x, sig, lam), "H")

    def lag_grad(z):
        return sig * P.eval_grad_f(z) + _dense(lay, P.eval_jac_g(z), "J").T @ lam
    Hfd = np.zeros_like(H)
    for j in range(lay.n):
        e = np.zeros(lay.n); e[j] = h
        Hfd[:, j] = (lag_grad(x + e) - lag_grad(x - e)) / (2 * h)
    assert np.abs(H - Hfd).max() < 1e-5 * max(1.0, np.abs(H).max())
    # every row is quadratic: the Hessian does not depend on the point
    assert np.array_equal(P.eval_h(x, sig, lam), P.eval_h(lay.x0, sig, lam))


def test_acr_and_polar_reach_the_same_optimum():
    """Oracle SQP-TR on both formulations of one network (textbook Hessian sign): same objective, same voltage
    magnitudes and dispatch (no angle limit binds at the optimum, so the feasible sets coincide)."""
    net = _net()
    nb, ng = net.nb, net.ng
    kw = dict(max_iter=60, tol_infeas=1e-6, tol_residual=1e-4, literal_quirks=0, use_soc=1)
    rp = O.sqp_solve(O.problem_acopf(net, acopf_layout(net)), O.default_options(**kw))
    rr = O.sqp_solve(O.problem_acopf(net, acr_layout(net)), O.default_options(**kw))
    assert rp["status"] == rr["status"] == 0
    assert abs(rp["obj_val"] - rr["obj_val"]) <= 1e-6 * abs(rp["obj_val"])
    vm = np.hypot(rr["x"][:nb], rr["x"][nb:2 * nb])
    assert np.abs(vm - rp["x"][nb:2 * nb]).max() < 1e-4
    assert np.abs(rr["x"][2 * nb:2 * nb + ng] - rp["x"][2 * nb:2 * nb + ng]).max() < 1e-4
    assert abs(rr["x"][net.ref_bus]) < 1e-9                         # vi[ref] = 0


@pytest.mark.parametrize("case", ["case14", "case118", "case1354"])
def test_symbolic_analysis_of_the_acr_structure(case):
    """The host-side analysis (no GPU) on the ACR Newton matrix: rows behind their variables, fronts and fill of the
    same order as the polar structure's -- the sparse path takes the formulation as it comes."""
    nb, ng, nl, seed = CASES[case]
    net = acopf_synth(nb, ng, nl, seed)
    lay, lp = acr_layout(net), acopf_layout(net)
    pos, st = pkg.kkt_symbolic(lay.n, lay.m, lay.jrow, lay.jcol, lay.hrow, lay.hcol, lay.gL, lay.gU)
    kept = np.flatnonzero(lay.gL == lay.gU)
    assert st["order"] == lay.n + len(kept) and sorted(pos.tolist()) == list(range(st["order"]))
    kpos = {int(i): k for k, i in enumerate(kept)}
    for r, c in zip(lay.jrow - 1, lay.jcol - 1):
        if int(r) in kpos:
            assert pos[lay.n + kpos[int(r)]] > pos[c]
    _, sp = pkg.kkt_symbolic(lp.n, lp.m, lp.jrow, lp.jcol, lp.hrow, lp.hcol, lp.gL, lp.gU)
    assert st["max_front"] <= 256 and st["nnz_l"] <= 1.5 * sp["nnz_l"] and st["flops"] <= 2.0 * sp["flops"]


# ------------------------------------------------------------------ W-space form (examples/acopf/acwr.jl)
def _lift(lay, va, vm, tail):
    """(vi, vr, w, wr, wi, tail) of the polar point (va, vm)."""
    nb = len(va)
    vr, vi = vm * np.cos(va), vm * np.sin(va)
    i, j = lay.bp_i, lay.bp_j
    return np.concatenate([vi, vr, vm ** 2, vr[i] * vr[j] + vi[i] * vi[j], vi[i] * vr[j] - vr[i] * vi[j], tail])


@pytest.mark.parametrize("shunts", [False, True])
def test_acwr_layout_counts_and_polar_equivalence(shunts):
    """Structure counts as sqphip_acopf_attach_acwr checks them; at a lifted polar point the model-voltage rows vanish
    and the Ohm and balance rows have the polar values (shunts included: gs w_i = gs vm_i^2)."""
    net = _net(shunts=shunts)
    nb, ng, nl = net.nb, net.ng, net.nl
    lp, lw = acopf_layout(net), acwr_layout(net)
    nbp = len(lw.bp_i)
    assert lw.form == "acwr" and lw.n == 3 * nb + 2 * nbp + 2 * ng + 4 * nl and lw.m == 1 + 3 * nb + 4 * nbp + 6 * nl
    assert len(lw.jrow) == 1 + 2 * (2 * nl + ng) + 2 * nb + 4 * nbp + 16 * nl + 3 * nb + 10 * nbp + 4 * nl
    assert len(lw.hrow) == ng + 4 * nl + 2 * nb + 4 * nbp and (lw.hrow >= lw.hcol).all()
    assert lw.num_linear == 1 + 2 * nb + 2 * nbp + 4 * nl           # everything but model voltage and thermal limits
    Pp, Pw = O.problem_acopf(net, lp), O.problem_acopf(net, lw)
    rng = np.random.default_rng(0)
    xp = lp.x0 + 0.05 * rng.standard_normal(lp.n)
    xw = _lift(lw, xp[:nb], xp[nb:2 * nb], xp[2 * nb:])
    gp, gw = Pp.eval_g(xp), Pw.eval_g(xw)
    O0p, O0w = 2 * nl + 1 + 2 * nb + 2 * nl, 1 + 2 * nb + 2 * nbp
    V0 = O0w + 4 * nl
    assert np.abs(gp[O0p:O0p + 4 * nl] - gw[O0w:O0w + 4 * nl]).max() < 1e-13
    assert np.abs(gp[2 * nl + 1:2 * nl + 1 + 2 * nb] - gw[1:1 + 2 * nb]).max() < 1e-13
    assert np.abs(gw[V0:V0 + nb + 2 * nbp]).max() < 1e-14
    # the angle rows are the polar angle limits in tangent form: same sign pattern of the slack
    th = xp[net.f_bus] - xp[net.t_bus]
    up = gw[1 + 2 * nb:O0w:2][lw.br_bp]; lo = gw[2 + 2 * nb:O0w:2][lw.br_bp]
    assert ((up <= 0) == (th * lw.br_sig <= np.arctan(lw.bp_tmax[lw.br_bp]) + 1e-12)).all()
    assert ((lo >= 0) == (th * lw.br_sig >= np.arctan(lw.bp_tmin[lw.br_bp]) - 1e-12)).all()


def test_acwr_derivatives_against_finite_differences():
    net = _net(shunts=True)
    lay = acwr_layout(net)
    P = O.problem_acopf(net, lay)
    rng = np.random.default_rng(1)
    x = lay.x0 + 0.05 * rng.standard_normal(lay.n); lam = rng.standard_normal(lay.m); sig = 0.7
    h = 1e-6
    J = _dense(lay, P.eval_jac_g(x), "J")
    Jfd = np.zeros_like(J)
    for j in range(lay.n):
        e = np.zeros(lay.n); e[j] = h
        Jfd[:, j] = (P.eval_g(x + e) - P.eval_g(x - e)) / (2 * h)
    assert np.abs(J - Jfd).max() < 1e-7
    H = _dense(lay, P.eval_h(x, sig, lam), "H")

    def lag_grad(z):
        return sig * P.eval_grad_f(z) + _dense(lay, P.eval_jac_g(z), "J").T @ lam
    Hfd = np.zeros_like(H)
    for j in range(lay.n):
        e = np.zeros(lay.n); e[j] = h
        Hfd[:, j] = (lag_grad(x + e) - lag_grad(x - e)) / (2 * h)
    assert np.abs(H - Hfd).max() < 1e-5 * max(1.0, np.abs(H).max())
    # the first num_linear rows are linear: their Jacobian entries do not depend on the point
    lin = lay.jrow <= lay.num_linear
    assert np.array_equal(P.eval_jac_g(x)[lin], P.eval_jac_g(lay.x0)[lin])


def test_acwr_reaches_the_polar_optimum():
    """The W-space model carries the polar model's constraints (voltage and angle limits included): same optimum."""
    net = _net()
    nb, ng = net.nb, net.ng
    kw = dict(max_iter=100, tol_infeas=1e-6, tol_residual=1e-4, literal_quirks=0, use_soc=1)
    rp = O.sqp_solve(O.problem_acopf(net, acopf_layout(net)), O.default_options(**kw))
    lw = acwr_layout(net)
    rw = O.sqp_solve(O.problem_acopf(net, lw), O.default_options(**kw))
    assert rp["status"] == rw["status"] == 0
    assert abs(rp["obj_val"] - rw["obj_val"]) <= 1e-6 * abs(rp["obj_val"])
    assert np.abs(np.sqrt(rw["x"][2 * nb:3 * nb]) - rp["x"][nb:2 * nb]).max() < 1e-4          # sqrt(w) = vm
    PG = 3 * nb + 2 * len(lw.bp_i)
    assert np.abs(rw["x"][PG:PG + ng] - rp["x"][2 * nb:2 * nb + ng]).max() < 1e-4
