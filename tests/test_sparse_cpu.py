"""CPU tests of the sparse path's host side (no GPU): the symbolic analysis of the Newton matrix, the multifrontal
plan (through the library's host reference of its numeric phase) and the oracle's independent sparse LDL^T.
The device kernels that run the same plan are tested under -m gpu (test_gpu_parity.py::test_multifrontal_*)."""
import ctypes as C

import numpy as np
import pytest

import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, contingency, renumber_buses, CASES
from oracle import oracle as O


def _layout(case, renumber=None):
    nb, ng, nl, seed = CASES[case]
    net = acopf_synth(nb, ng, nl, seed)
    if renumber is not None:
        net = renumber_buses(net, renumber)
    return net, acopf_layout(net)


def _dense_newton(lay, cond, Jv, Hv, Dd, sigp, hd, rt, hsc, dw):
    """The matrix sqphip_mf_host_solve / the kernels factorise, assembled densely by an independent route."""
    n, m = lay.n, lay.m
    J = np.zeros((m, n)); np.add.at(J, (lay.jrow - 1, lay.jcol - 1), Jv)
    H = np.zeros((n, n)); np.add.at(H, (lay.hrow - 1, lay.hcol - 1), Hv); H = H + H.T - np.diag(np.diag(H))
    cnt = np.bincount(lay.jrow - 1, minlength=m)
    kept = ((lay.gL == lay.gU) | (cnt > 32)) if cond else np.ones(m, bool)
    J = J * (rt != 0)[:, None]
    W = hsc * H + np.diag(hd + sigp + dw + 1e-8)
    el = ~kept & (rt != 0)
    W = W + J[el].T @ (J[el] / (Dd[el] + 1e-8)[:, None])
    Dk = np.where(rt[kept] != 0, Dd[kept] + 1e-8, 1.0)
    return np.block([[W, J[kept].T], [J[kept], -np.diag(Dk)]]), kept


def _random_values(lay, seed, free_frac=0.0):
    rng = np.random.default_rng(seed)
    eq = lay.gL == lay.gU
    Dd = rng.uniform(0.1, 10, lay.m); Dd[eq] = rng.uniform(0, 1e-3, eq.sum())
    rt = np.ones(lay.m, dtype=np.int32); rt[rng.uniform(size=lay.m) < free_frac] = 0
    return (rng.normal(size=len(lay.jrow)), 0.1 * rng.normal(size=len(lay.hrow)), Dd, rng.uniform(1, 20, lay.n),
            rng.uniform(0, 1, lay.n), rt)


@pytest.mark.parametrize("case,cond,free", [("case14", 1, 0.0), ("case14", 0, 0.2), ("case118", 1, 0.0), ("case118", 1, 0.3),
                                            ("case118", 0, 0.0)])
def test_multifrontal_plan_reproduces_a_dense_solve(case, cond, free, monkeypatch):
    """Plan = ordering + supernodes + assembly lists + extend-add maps.  Its host reference (front-by-front partial
    LDL^T with the right-hand side carried along) must solve the Newton system like numpy does, with the right inertia."""
    monkeypatch.setenv("SQPHIP_MF_SPINE", "1")      # the spine kernel's plan is built on request only (mfplan.hip)
    _, lay = _layout(case)
    Jv, Hv, Dd, sigp, hd, rt = _random_values(lay, 3, free)
    K, kept = _dense_newton(lay, cond, Jv, Hv, Dd, sigp, hd, rt, 0.7, 1e-3)
    rhs = np.random.default_rng(5).normal(size=K.shape[0])
    sol, dinv, npos = pkg.mf_host_solve(lay.n, lay.m, lay.jrow, lay.jcol, lay.hrow, lay.hcol, lay.gL, lay.gU, cond, Jv, Hv,
                                        Dd, sigp, hd, rt, 0.7, 1e-3, rhs)
    ref = np.linalg.solve(K, rhs)
    assert np.abs(sol - ref).max() <= 1e-11 * np.abs(ref).max()
    # the streamed top-of-tree solve (k_mf_solve_top2) replayed on the host from its own plan arrays (front records,
    # gather lists over the vector of updates, row maps) must reproduce the plain recursion; -1: no such top (the full
    # forms end in fronts beyond the kernel's 84 columns / 128 rows: 96 x 0 for IEEE-14, 162 columns for IEEE-118)
    err = pkg.mf_host_top2_err()
    assert (err == -1.0 and cond == 0) or 0.0 <= err <= 1e-12, err
    # ... and the front assembly of the spine kernel (k_mf_spine) from ITS arrays: gather entries whose sources lie in the
    # arena, the block a front hands to its parent through the row map, the destination list -- against the images the plain
    # recursion assembled; IEEE-14 has no narrow top, the full forms end in fronts of more than eight tiles
    serr = pkg.mf_host_spine_err()
    assert (serr == -1.0 and (cond == 0 or case == "case14")) or 0.0 <= serr <= 1e-13, serr
    assert npos == lay.n == int((np.linalg.eigvalsh(K) > 0).sum())
    assert np.all(np.isfinite(dinv)) and int((dinv > 0).sum()) == lay.n


@pytest.mark.parametrize("case,renumber", [("case14", None), ("case118", None), ("case118", 7), ("case1354", None)])
def test_symbolic_analysis_invariants(case, renumber):
    """The order is a permutation; every row of the matrix sits behind every variable it couples to (the rule that
    keeps the pivots of the quasi-definite matrix away from the 1e-8 regularisation); renumbering the buses at random
    changes nothing essential (no dependence on the synthetic numbering); the numbers the bench reports are sane."""
    _, lay = _layout(case, renumber)
    pos, st = pkg.kkt_symbolic(lay.n, lay.m, lay.jrow, lay.jcol, lay.hrow, lay.hcol, lay.gL, lay.gU)
    nu = st["order"]
    assert sorted(pos.tolist()) == list(range(nu))
    kept = np.flatnonzero(lay.gL == lay.gU)
    assert nu == lay.n + len(kept)
    kpos = {int(i): k for k, i in enumerate(kept)}
    for r, c in zip(lay.jrow - 1, lay.jcol - 1):
        if int(r) in kpos:
            assert pos[lay.n + kpos[int(r)]] > pos[c]
    assert st["nnz_l"] >= st["nnz_l_exact"] > 0 and st["flops"] >= st["flops_exact"] * 0.99
    assert st["max_front"] <= 256 and st["n_levels"] <= 80
    if renumber is not None:
        _, lay0 = _layout(case)
        _, st0 = pkg.kkt_symbolic(lay0.n, lay0.m, lay0.jrow, lay0.jcol, lay0.hrow, lay0.hcol, lay0.gL, lay0.gU)
        assert st["nnz_l_exact"] <= 1.3 * st0["nnz_l_exact"] and st["max_front"] <= 1.5 * st0["max_front"]


def test_rows_may_not_precede_their_variables_costs_fill_not_correctness():
    """The unconstrained minimum-degree order has a third of the fill (rows_after_vars = 0); it is available for
    experiments only -- the constrained one is what the library uses."""
    _, lay = _layout("case118")
    _, a = pkg.kkt_symbolic(lay.n, lay.m, lay.jrow, lay.jcol, lay.hrow, lay.hcol, lay.gL, lay.gU, rows_after_vars=True)
    _, b = pkg.kkt_symbolic(lay.n, lay.m, lay.jrow, lay.jcol, lay.hrow, lay.hcol, lay.gL, lay.gU, rows_after_vars=False)
    assert b["nnz_l_exact"] < a["nnz_l_exact"]


def _structure_with_a_dense_row(n=100):
    """n variables, a banded Hessian, a few short equality / inequality rows and ONE inequality over all variables."""
    jr, jc = [], []
    for i in range(10):                       # short rows
        for j in (3 * i, 3 * i + 1, 3 * i + 2):
            jr.append(i + 1); jc.append(j + 1)
    for j in range(n):                        # sum(x) <= b
        jr.append(11); jc.append(j + 1)
    hr = list(range(1, n + 1)) + list(range(2, n + 1)); hc = list(range(1, n + 1)) + list(range(1, n))
    gL = np.array([0.0] * 5 + [-np.inf] * 5 + [-np.inf]); gU = np.array([0.0] * 5 + [1.0] * 5 + [10.0])
    return n, 11, np.array(jr, dtype=np.int64), np.array(jc, dtype=np.int64), np.array(hr, dtype=np.int64), \
        np.array(hc, dtype=np.int64), gL, gU


def test_a_long_inequality_row_stays_in_the_matrix():
    """a row with more than 32 entries used to enter the ordering graph as a chain although its
    elimination creates a clique, and coupled variables ended up in different 'independent' leading tiles.  Such rows
    now stay in the matrix (kkt_row_is_kept): the condensed order counts them, no leading tile of the dense tile order
    is coupled to another, and the sparse plan solves the system."""
    n, m, jr, jc, hr, hc, gL, gU = _structure_with_a_dense_row()
    pos, ts, nf = pkg.kkt_order(n, m, jr, jc, hr, hc, gL, gU)
    assert len(pos) == n + 6                                   # 5 equalities + the long row
    tile = lambda u: pos[u] // 64 if pos[u] < 64 * ts else -1  # leading tile of an unknown, -1 = dense remainder
    coupled = [(int(a) - 1, int(b) - 1) for a, b in zip(hr, hc) if a != b]
    rows = {}
    for r, c in zip(jr, jc):
        rows.setdefault(int(r) - 1, []).append(int(c) - 1)
    for i, cols in rows.items():
        if gL[i] != gU[i] and len(cols) <= 32:                 # eliminated row: a clique among its variables
            coupled += [(a, b) for a in cols for b in cols if a < b]
    for a, b in coupled:
        assert tile(a) == tile(b) or tile(a) < 0 or tile(b) < 0, (a, b)
    # the sparse plan on the same structure
    rng = np.random.default_rng(1)
    Jv = rng.normal(size=len(jr)); Hv = np.concatenate([rng.uniform(2, 3, n), 0.3 * rng.normal(size=n - 1)])
    Dd = np.where(gL == gU, 1e-6, rng.uniform(0.5, 2, m)); sigp = rng.uniform(1, 2, n); hd = np.zeros(n)
    rt = np.ones(m, dtype=np.int32)
    rhs = rng.normal(size=n + 6)
    sol, dinv, npos = pkg.mf_host_solve(n, m, jr, jc, hr, hc, gL, gU, 1, Jv, Hv, Dd, sigp, hd, rt, 1.0, 0.0, rhs)

    class L_:                                                  # the layout fields _dense_newton reads
        pass
    lay = L_(); lay.n, lay.m, lay.jrow, lay.jcol, lay.hrow, lay.hcol, lay.gL, lay.gU = n, m, jr, jc, hr, hc, gL, gU
    K, kept = _dense_newton(lay, 1, Jv, Hv, Dd, sigp, hd, rt, 1.0, 0.0)
    assert kept.sum() == 6 and kept[10]
    ref = np.linalg.solve(K, rhs)
    assert np.abs(sol - ref).max() <= 1e-11 * np.abs(ref).max() and npos == n


# ------------------------------------------------------------------ the oracle's own sparse LDL^T
def test_oracle_sparse_ldlt_against_numpy():
    L = O.lib()
    lp, dp = C.POINTER(C.c_int64), C.POINTER(C.c_double)
    L.ora_sldl_analyse.restype = C.c_void_p
    L.ora_sldl_analyse.argtypes = [C.c_int64, C.c_int64, lp, lp, lp, lp, C.c_int]
    L.ora_sldl_numeric.restype = C.c_int64
    L.ora_sldl_numeric.argtypes = [C.c_void_p, dp, lp]
    L.ora_sldl_solve.argtypes = [C.c_void_p, dp]
    L.ora_sldl_free.argtypes = [C.c_void_p]
    rng = np.random.default_rng(2)
    n1, n2 = 60, 40
    n = n1 + n2
    A = np.zeros((n, n))
    for _ in range(260):
        i, j = rng.integers(0, n, 2)
        if i != j:
            A[max(i, j), min(i, j)] = rng.normal()
    A = A + A.T + np.diag(np.concatenate([rng.uniform(8, 12, n1), -rng.uniform(8, 12, n2)]))
    ti, tj = np.nonzero(np.tril(A))
    tv = A[ti, tj].copy()
    ti = np.concatenate([ti, ti[:7]]).astype(np.int64); tj = np.concatenate([tj, tj[:7]]).astype(np.int64)
    tv = np.concatenate([tv, np.zeros(7)]); tv[-7:] = 0.5 * tv[:7]; tv[:7] *= 0.5      # duplicates are summed
    for natural in (0, 1):
        S = L.ora_sldl_analyse(n, len(ti), ti.ctypes.data_as(lp), tj.ctypes.data_as(lp), None, None, natural)
        bad = C.c_int64()
        npos = L.ora_sldl_numeric(S, tv.ctypes.data_as(dp), C.byref(bad))
        assert npos == n1 and bad.value == 0
        b = rng.normal(size=n); x = b.copy()
        L.ora_sldl_solve(S, x.ctypes.data_as(dp))
        assert np.abs(A @ x - b).max() <= 1e-12 * np.abs(b).max() * np.linalg.cond(A)
        L.ora_sldl_free(S)


@pytest.mark.parametrize("quirks", [1, 0])
def test_oracle_sparse_and_dense_linear_algebra_agree_on_a_whole_solve(quirks):
    """Same interior-point method, two factorisations (dense in natural order, sparse in the oracle's minimum-degree
    order): a 14-bus SQP run to convergence gives the same decisions, counts and optimum."""
    nb, ng, nl, seed = CASES["case14"]
    base = acopf_synth(nb, ng, nl, seed)
    for net in (base, contingency(base, 5, seed)):
        lay = acopf_layout(net)
        kw = dict(max_iter=40, tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, literal_quirks=quirks)
        rd = O.sqp_solve(O.problem_acopf(net, lay), O.default_options(kkt_mode=1, **kw))
        rs = O.sqp_solve(O.problem_acopf(net, lay), O.default_options(kkt_mode=2, **kw))
        assert (rd["status"], rd["iter"], rd["n_qp"]) == (rs["status"], rs["iter"], rs["n_qp"])
        assert abs(rd["n_ipm_iter"] - rs["n_ipm_iter"]) <= max(2, 0.02 * rd["n_ipm_iter"])
        tol = 1e-8 if rd["status"] == 0 else 1e-5
        assert np.abs(rd["x"] - rs["x"]).max() <= tol * max(1.0, np.abs(rd["x"]).max())


def test_oracle_gives_a_cycling_correction_half_the_iteration_limit():
    """tests/golden/soc_cycling_subproblem.npz (a second-order correction of the IEEE-118 workload on which the
    regularised Newton iteration cycles; fetched from the device, see tests/test_gpu_parity.py): as a correction the
    oracle stops at half of options.ipm_max_iter, as an ordinary sub-problem at the whole limit, both ITERATION_LIMIT --
    the rule the product follows (include/sqphip.h, options.ipm_max_iter)."""
    import os
    from sqpsolver_jl_amd.acopf_synth import contingency
    D = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "soc_cycling_subproblem.npz"))
    nb, ng, nl, seed = CASES["case118"]
    lay = acopf_layout(contingency(acopf_synth(nb, ng, nl, seed), 109, seed))
    jcp, jrv, jslot, _ = O.coo_to_csc(lay.n, lay.jrow, lay.jcol)
    hcp, hrv, hslot, hslot_t = O.coo_to_csc(lay.n, lay.hrow, lay.hcol, sym=True)
    jv = np.zeros(len(jrv)); np.add.at(jv, jslot, D["jac_coo"])
    hv = np.zeros(len(hrv)); np.add.at(hv, hslot, D["hess_coo"]); ok = hslot_t >= 0; np.add.at(hv, hslot_t[ok], D["hess_coo"][ok])
    o = O.default_options(tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, literal_quirks=1, num_threads=4, kkt_mode=2)
    q = O.QpSolver(lay.n, lay.m, lay.num_linear, jcp, jrv, hcp, hrv, lay.xL, lay.xU, lay.gL, lay.gU, o)
    for mode, limit in ((2, o.ipm_max_iter // 2), (0, o.ipm_max_iter)):
        r = q.solve(mode, D["x_k"], float(D["delta"]), float(D["mu_pen"]), D["c"], D["b"], jv, hv)
        assert (r["status"], r["ipm_iters"]) == (11, limit)
