"""MATPOWER reader: the reference's on-disk input format
(/root/reference/examples/acopf/case3.m:7-36, loaded at examples/acopf/opf.jl:12-16)."""
import os

import numpy as np
import pytest

import sqpsolver_jl_amd as pkg   # noqa: F401  (registers the package)
from sqpsolver_jl_amd import matpower as MP
from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout

HAND = """
% hand-written 4-bus case: non-contiguous bus numbers, comments, mixed row separators
function mpc = case4_hand
mpc.version = '2';
mpc.baseMVA = 50.0;   % not 100 on purpose
mpc.note = 'a % sign inside a string is not a comment';
mpc.bus = [
    10  3  0.0   0.0   0 0 1 1.0 0.0 230 1 1.05 0.95;
    20  1  40.0  10.0  0 0 1 1.0 0.0 230 1 1.05 0.95; % load bus
    35  2  25.0  5.0   0 0 1 1.0 0.0 230 1 1.06 0.94
    7   1  30.0  7.5   0 0 1 1.0 0.0 230 1 1.05 0.95;
];
mpc.gen = [
    10  0 0  60 -60  1.0 50 1  120 0;
    35  0 0  30 -30  1.0 50 1  80  10;
    20  0 0  30 -30  1.0 50 0  80  0;   % out of service
];
mpc.gencost = [
    2 0 0 3  0.02 12.0 100.0;
    2 0 0 2  20.0 0.0  0.0;
    2 0 0 3  0.05 30.0 0.0;
];
mpc.branch = [
    10 20 0.01 0.10 0.02  100 0 0 0 0 1 -30 30;
    20 35 0.02 0.15 0.03  0   0 0 1 0 1 0 0;
    35 7  0.01 0.12 0.02  80  0 0 0 0 0 -30 30;
    7  10 0.015 0.11 0.01 90  0 0 0 0 1 -360 360;
];
mpc.bus_name = { 'a'; 'b'; 'c'; 'd'; };
"""


def test_parse_fields_comments_and_strings():
    m = MP.read_matpower(HAND)
    assert m["baseMVA"] == 50.0 and m["version"] == "2"
    assert m["note"] == "a % sign inside a string is not a comment"
    assert m["bus"].shape == (4, 13) and m["gen"].shape == (3, 10) and m["branch"].shape == (4, 13)
    assert m["gencost"].shape == (3, 7)
    assert "bus_name" not in m                      # cell arrays are skipped
    with pytest.raises(ValueError):
        MP.read_matpower("mpc.bus = [1 2 3; 4 5];\nmpc.gen=[1];\nmpc.branch=[1];")   # ragged
    with pytest.raises(ValueError):
        MP.read_matpower("function mpc = x\nmpc.baseMVA = 100;\n")                     # no matrices


def test_conversion_per_unit_renumbering_status():
    net = MP.network_from_matpower(MP.read_matpower(HAND))
    assert (net.nb, net.ng, net.nl) == (4, 2, 4)            # the out-of-service generator is dropped
    assert net.ref_bus == 0 and net.gen_bus.tolist() == [0, 2]
    assert np.allclose(net.pd, [0.0, 0.8, 0.5, 0.6]) and np.allclose(net.qd, [0, 0.2, 0.1, 0.15])
    assert net.f_bus.tolist() == [0, 1, 2, 3] and net.t_bus.tolist() == [1, 2, 3, 0]
    assert net.status.tolist() == [1.0, 1.0, 0.0, 1.0]      # open branch keeps its slot
    assert np.allclose(net.pmax, [2.4, 1.6]) and np.allclose(net.pmin, [0.0, 0.2])
    assert np.allclose(net.qmax, [1.2, 0.6]) and np.allclose(net.qmin, [-1.2, -0.6])
    # cost in per-unit variables: c2*base^2, c1*base; a degree-1 polynomial has c2 = 0
    assert np.allclose(net.c2, [0.02 * 2500, 0.0]) and np.allclose(net.c1, [12.0 * 50, 20.0 * 50])
    assert np.allclose(net.rate_a, [2.0, 1e4, 1.6, 1.8])    # rateA = 0 -> unlimited
    # angle limits: 0/0 and beyond +-60 degrees fall back to +-60 degrees
    assert np.allclose(net.angmax, [np.pi / 6, np.pi / 3, np.pi / 6, np.pi / 3])
    assert np.allclose(net.vmin, [0.95, 0.95, 0.94, 0.95]) and np.allclose(net.vmax, [1.05, 1.05, 1.06, 1.05])
    lay = acopf_layout(net)
    assert lay.n == 2 * 4 + 2 * 2 + 4 * 4 and lay.m == 1 + 2 * 4 + 8 * 4


@pytest.mark.parametrize("edit,needle", [
    (("0.01 0.10 0.02  100 0 0 0 0 1 -30 30", "0.01 0.10 0.02  100 0 0 -0.98 0 1 -30 30"), "negative tap"),
    (("2 0 0 3  0.02 12.0 100.0", "1 0 0 3  0.02 12.0 100.0"), "piecewise"),
    (("mpc.bus_name", "mpc.dcline = [10 20 1 10 10 5 0 1 1 10 90 -90 90 -90 90 0 0 0 0 0 0 0 0];\n"
                      "mpc.dclinecost = [2 0 0 2 1.0 0.0];\nmpc.bus_name"), "HVDC line costs"),
])
def test_unsupported_features_are_rejected_loudly(edit, needle):
    txt = HAND.replace(*edit)
    assert txt != HAND
    with pytest.raises(MP.UnsupportedCase, match=needle):
        MP.network_from_matpower(MP.read_matpower(txt))
    if needle.startswith("HVDC"):
        assert MP.network_from_matpower(MP.read_matpower(txt), dcline="drop").ndc == 0


def test_taps_and_phase_shifters_are_read_and_reach_the_flow_equations():
    """ratio / angle columns of mpc.branch -> Network.tap / shift (radians) -> the twelve Ohm's-law coefficients per
    branch, checked against the complex-power flows of MATPOWER's branch admittance matrix
        [I_f; I_t] = [[(y + j bc/2)/tau^2, -y/(tau e^{-j phi})], [-y/(tau e^{j phi}), y + j bc/2]] [V_f; V_t]."""
    txt = HAND.replace("0.01 0.10 0.02  100 0 0 0 0 1 -30 30", "0.01 0.10 0.02  100 0 0 0.97 4.0 1 -30 30")
    assert txt != HAND
    net = MP.network_from_matpower(MP.read_matpower(txt))
    l = int(np.flatnonzero(net.tap != 1.0)[0])
    assert net.tap[l] == 0.97 and np.isclose(net.shift[l], np.deg2rad(4.0))
    assert (np.delete(net.tap, l) == 1.0).all() and (np.delete(net.shift, l) == 0.0).all()
    rng = np.random.default_rng(0)
    net.tap = rng.uniform(0.9, 1.1, net.nl); net.shift = rng.uniform(-0.2, 0.2, net.nl)
    co = net.branch_coeffs()
    va = rng.uniform(-0.3, 0.3, net.nb); vm = rng.uniform(0.9, 1.1, net.nb)
    V = vm * np.exp(1j * va)
    y = 1.0 / (net.r + 1j * net.x)
    for b in range(net.nl):
        f, t = net.f_bus[b], net.t_bus[b]
        tau, phi = net.tap[b], net.shift[b]
        i_f = (y[b] + 0.5j * net.bc[b]) / tau ** 2 * V[f] - y[b] / (tau * np.exp(-1j * phi)) * V[t]
        i_t = -y[b] / (tau * np.exp(1j * phi)) * V[f] + (y[b] + 0.5j * net.bc[b]) * V[t]
        s_f, s_t = V[f] * np.conj(i_f) * net.status[b], V[t] * np.conj(i_t) * net.status[b]
        th, uu = va[f] - va[t], vm[f] * vm[t]
        F = [co[b, 3 * k] * (vm[t] if k >= 2 else vm[f]) ** 2 + uu * (co[b, 3 * k + 1] * np.cos(th) + co[b, 3 * k + 2] * np.sin(th))
             for k in range(4)]
        assert np.allclose(F, [s_f.real, s_f.imag, s_t.real, s_t.imag], rtol=1e-13, atol=1e-14)
    back = MP.load_case(MP.write_matpower(net, "tapped"))
    assert np.allclose(back.tap, net.tap, rtol=1e-15) and np.allclose(back.shift, net.shift, rtol=1e-13, atol=1e-16)


def test_bus_shunts_are_read_and_enter_the_balance_rows():
    """Gs / Bs columns of mpc.bus -> Network.gs / bs (per unit) -> vm^2 terms of the balance rows, checked against
    the complex bus injection  S_i = sum of branch flows + V_i conj(Y_sh V_i),  Y_sh = gs + j bs; the rows become
    nonlinear rows and the structure gains 2 Jacobian + 1 Hessian entries per shunted bus."""
    from oracle import oracle as O
    txt = HAND.replace("20  1  40.0  10.0  0 0", "20  1  40.0  10.0  1.5 4.5")
    assert txt != HAND
    net = MP.network_from_matpower(MP.read_matpower(txt))
    sb, gs, bs = net.shunts()
    assert sb.tolist() == [1] and np.allclose(gs, [0.03]) and np.allclose(bs, [0.09])      # baseMVA = 50
    plain = MP.network_from_matpower(MP.read_matpower(HAND))
    lay, lay0 = acopf_layout(net), acopf_layout(plain)
    assert lay.num_linear == 2 * net.nl + 1 and lay0.num_linear == 2 * net.nl + 1 + 2 * net.nb
    assert len(lay.jrow) == len(lay0.jrow) + 2 and len(lay.hrow) == len(lay0.hrow) + 1
    rng = np.random.default_rng(5)
    x = np.clip(lay.x0 + 0.05 * rng.standard_normal(lay.n), lay.xL, lay.xU)
    g, g0 = O.problem_acopf(net, lay).eval_g(x), O.problem_acopf(plain, lay0).eval_g(x)
    vm = x[net.nb + 1]
    d = g - g0
    rp, rq = 2 * net.nl + 1 + 2 * 1, 2 * net.nl + 2 + 2 * 1
    assert np.isclose(d[rp], 0.03 * vm ** 2, rtol=1e-13) and np.isclose(d[rq], -0.09 * vm ** 2, rtol=1e-13)
    assert np.count_nonzero(d) == 2
    s_sh = vm ** 2 * np.conj(0.03 + 0.09j)                      # V conj(Y V) = |V|^2 conj(Y)
    assert np.isclose(d[rp], s_sh.real) and np.isclose(d[rq], s_sh.imag)
    back = MP.load_case(MP.write_matpower(net, "shunted", base_mva=50.0))
    assert np.allclose(back.gs, net.gs) and np.allclose(back.bs, net.bs)


@pytest.mark.parametrize("shunts", [False, True])
def test_tapped_network_evaluator_derivatives_match_finite_differences(shunts):
    """The oracle's ACOPF callbacks on a network with random taps and shifts: Jacobian and Hessian of the
    Lagrangian against central differences of eval_g / eval_jac_g."""
    from oracle import oracle as O
    import scipy.sparse as sp
    net = acopf_synth(14, 5, 20, 14)
    rng = np.random.default_rng(3)
    net.tap = rng.uniform(0.92, 1.08, net.nl); net.shift = rng.uniform(-0.1, 0.1, net.nl)
    if shunts:
        net.gs = np.where(rng.random(net.nb) < 0.3, rng.uniform(0.0, 0.05, net.nb), 0.0)
        net.bs = np.where(rng.random(net.nb) < 0.3, rng.uniform(-0.1, 0.2, net.nb), 0.0)
    lay = acopf_layout(net)
    assert (len(lay.sh_bus) > 0) == shunts
    P = O.problem_acopf(net, lay)
    x = np.clip(lay.x0 + 0.05 * rng.standard_normal(lay.n), lay.xL, lay.xU)
    lam = rng.standard_normal(lay.m)
    J = sp.coo_matrix((P.eval_jac_g(x), (lay.jrow - 1, lay.jcol - 1)), shape=(lay.m, lay.n)).toarray()
    Hl = sp.coo_matrix((P.eval_h(x, 0.7, lam), (lay.hrow - 1, lay.hcol - 1)), shape=(lay.n, lay.n)).toarray()
    H = Hl + Hl.T - np.diag(np.diag(Hl))
    h = 1e-6
    Jfd = np.zeros_like(J); Hfd = np.zeros_like(H)
    for j in range(lay.n):
        e = np.zeros(lay.n); e[j] = h
        Jfd[:, j] = (P.eval_g(x + e) - P.eval_g(x - e)) / (2 * h)
        def lag_grad(z):
            Jz = sp.coo_matrix((P.eval_jac_g(z), (lay.jrow - 1, lay.jcol - 1)), shape=(lay.m, lay.n)).toarray()
            return 0.7 * P.eval_grad_f(z) + Jz.T @ lam
        Hfd[:, j] = (lag_grad(x + e) - lag_grad(x - e)) / (2 * h)
    assert np.abs(J - Jfd).max() < 1e-7 * max(1.0, np.abs(J).max())
    assert np.abs(H - Hfd).max() < 1e-6 * max(1.0, np.abs(H).max())


def test_write_read_round_trip_of_a_synthetic_case():
    net = acopf_synth(14, 5, 20, 14)
    back = MP.load_case(MP.write_matpower(net, "case14_synth"))
    for fld in ("pd", "qd", "vmin", "vmax", "pmin", "pmax", "qmin", "qmax", "c2", "c1", "r", "x", "bc", "rate_a",
                "angmin", "angmax", "status"):
        assert np.allclose(getattr(net, fld), getattr(back, fld), rtol=1e-13, atol=1e-15), fld
    assert back.ref_bus == net.ref_bus and back.gen_bus.tolist() == net.gen_bus.tolist()
    assert back.f_bus.tolist() == net.f_bus.tolist() and back.t_bus.tolist() == net.t_bus.tolist()


def test_golden_case_file_solves_like_the_generator_output():
    """tests/golden/case14_synth.m (written by make_golden.py) -> same optimum as the in-memory network."""
    from oracle import oracle as O
    here = os.path.dirname(os.path.abspath(__file__))
    net = MP.load_case(os.path.join(here, "golden", "case14_synth.m"))
    ref = acopf_synth(14, 5, 20, 14)
    opt = O.default_options(max_iter=40, tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, literal_quirks=0)
    a = O.sqp_solve(O.problem_acopf(net, acopf_layout(net)), opt)
    b = O.sqp_solve(O.problem_acopf(ref, acopf_layout(ref)), opt)
    assert a["status"] == b["status"] == 0 and a["iter"] == b["iter"]
    assert np.allclose(a["x"], b["x"], rtol=1e-7, atol=1e-9)


@pytest.mark.skipif(not os.path.exists("/root/reference/examples/acopf/case3.m"), reason="reference not mounted")
def test_reference_example_case_is_read_in_place():
    """The reference's own example file (read where it lies, never copied): 3 buses, 3 generators, 3 branches,
    one HVDC line that the polar evaluator does not model."""
    m = MP.read_matpower("/root/reference/examples/acopf/case3.m")
    assert m["baseMVA"] == 100.0 and m["bus"].shape == (3, 13) and m["dcline"].shape[0] == 1
    assert m["const_str"] == "a string" and m["const_int"] == 123.0
    net = MP.network_from_matpower(m)
    assert net.ndc == 1 and net.dcline["f_bus"].tolist() == [0] and net.dcline["t_bus"].tolist() == [1]
    assert np.allclose([net.dcline["pminf"][0], net.dcline["pmaxf"][0], net.dcline["loss0"][0], net.dcline["loss1"][0]],
                       [0.1, 9.0, 0.0, 0.0])
    assert MP.network_from_matpower(m, dcline="drop").ndc == 0
    assert (net.nb, net.ng, net.nl) == (3, 3, 3) and net.ref_bus == 0     # no type-3 bus in the file: largest generator's bus
    assert np.allclose(net.pd, [1.1, 1.1, 0.95]) and np.allclose(net.rate_a, [90.0, 0.5, 90.0])
    assert np.allclose(net.c2, [1100.0, 850.0, 0.0]) and np.allclose(net.c1, [500.0, 120.0, 0.0])


@pytest.mark.skipif(not os.path.exists("/root/reference/examples/acopf/case3.m"), reason="reference not mounted")
def test_reference_example_case_solves_to_the_dispatch_stored_in_the_file():
    """The reference's example file carries the solved ACOPF in its columns (gen Pg = 158.067, 160.006, 0 MW; dcline
    Pf = Pmin = 10 MW; PowerModels' documented objective for this case is 5907).  The full model -- AC branches plus
    the HVDC line with its loss row, examples/acopf/opf.jl:12-46 -- solved by the restated SQP-TR from a flat start
    must land on that dispatch, under both Hessian sign conventions."""
    from oracle import oracle as O
    m = MP.read_matpower("/root/reference/examples/acopf/case3.m")
    net = MP.network_from_matpower(m)
    lay = acopf_layout(net)
    assert lay.n == 2 * 3 + 2 * 3 + 4 * 3 + 4 and lay.m == 1 + 2 * 3 + 8 * 3 + 1
    for quirks in (0, 1):
        r = O.sqp_solve(O.problem_acopf(net, lay), O.default_options(max_iter=100, tol_infeas=1e-6, tol_residual=1e-4,
                                                                     use_soc=1, literal_quirks=quirks))
        assert r["status"] == 0
        pg = r["x"][2 * net.nb:2 * net.nb + net.ng] * m["baseMVA"]
        assert np.allclose(pg, m["gen"][:, 1], atol=2e-3)               # stored with 3 decimals (MW)
        assert abs(r["obj_val"] - 5906.88) < 0.01
        p_dc = r["x"][-4:-2] * m["baseMVA"]
        assert np.allclose(p_dc, [m["dcline"][0, 3], -m["dcline"][0, 4]], atol=1e-4)   # Pf = 10 MW in, Pt = 10 MW out


def test_dcline_loss_rows_and_balance_terms_match_finite_differences():
    from oracle import oracle as O
    import dataclasses
    import scipy.sparse as sp
    net = acopf_synth(14, 5, 20, 14)
    dc = dict(f_bus=np.array([2, 7], dtype=np.int32), t_bus=np.array([9, 3], dtype=np.int32),
              pminf=np.array([0.05, -0.3]), pmaxf=np.array([0.6, 0.3]), qminf=np.full(2, -0.4), qmaxf=np.full(2, 0.4),
              qmint=np.full(2, -0.4), qmaxt=np.full(2, 0.4), loss0=np.array([0.002, 0.0]), loss1=np.array([0.03, 0.0]))
    net = dataclasses.replace(net, dcline=dc)
    lay = acopf_layout(net)
    assert lay.n == 118 + 8 and lay.m == 189 + 2 and len(lay.jrow) == 651 + 2 * 6
    P = O.problem_acopf(net, lay)
    rng = np.random.default_rng(4)
    x = np.clip(lay.x0 + 0.05 * rng.standard_normal(lay.n), lay.xL, lay.xU)
    J = sp.coo_matrix((P.eval_jac_g(x), (lay.jrow - 1, lay.jcol - 1)), shape=(lay.m, lay.n)).toarray()
    h = 1e-6
    for j in range(lay.n):
        e = np.zeros(lay.n); e[j] = h
        assert np.abs((P.eval_g(x + e) - P.eval_g(x - e)) / (2 * h) - J[:, j]).max() < 1e-7 * max(1.0, np.abs(J).max())
    g = P.eval_g(x)
    DC = lay.n - 8
    assert np.allclose(g[-2:], [(1 - 0.03) * x[DC] + x[DC + 2], x[DC + 1] + x[DC + 3]])
    plain = acopf_synth(14, 5, 20, 14)
    g0 = O.problem_acopf(plain, acopf_layout(plain)).eval_g(x[:118])
    d = g[:189] - g0                                              # the dc terminals load the balance rows of their buses
    rowP = lambda b: 2 * 20 + 1 + 2 * b
    assert np.isclose(d[rowP(2)], x[DC]) and np.isclose(d[rowP(9)], x[DC + 2]) and np.isclose(d[rowP(2) + 1], x[DC + 4])
    assert np.count_nonzero(np.abs(d) > 1e-15) == 8
    # the whole thing solves
    r = O.sqp_solve(P, O.default_options(max_iter=60, tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, literal_quirks=0))
    assert r["status"] == 0


def test_golden_copy_of_the_reference_case_reproduces_its_stored_dispatch():
    """tests/golden/case3_network.m is the reference's example network re-serialised by make_golden.py (our writer's
    text, not the reference's file); case3_dispatch.json holds the dispatch that file stores.  Runs where
    /root/reference does not exist (the GPU box)."""
    import json
    from oracle import oracle as O
    here = os.path.dirname(os.path.abspath(__file__))
    pin = json.load(open(os.path.join(here, "golden", "case3_dispatch.json")))["case3_dispatch"]
    net = MP.load_case(os.path.join(here, "golden", "case3_network.m"))
    assert (net.nb, net.ng, net.nl, net.ndc) == (3, 3, 3, 1)
    lay = acopf_layout(net)
    r = O.sqp_solve(O.problem_acopf(net, lay), O.default_options(max_iter=100, tol_infeas=1e-6, tol_residual=1e-4,
                                                                 use_soc=1, literal_quirks=1))
    assert r["status"] == 0
    assert np.allclose(r["x"][2 * net.nb:2 * net.nb + net.ng] * pin["base_mva"], pin["pg_mw"], atol=2e-3)
    assert np.allclose(r["x"][-4:-2] * pin["base_mva"], [pin["dc_pf_mw"], -pin["dc_pt_mw"]], atol=1e-4)
    assert abs(r["obj_val"] - pin["objective"]) < 0.01
