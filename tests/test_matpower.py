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
    (("0.01 0.10 0.02  100 0 0 0 0 1 -30 30", "0.01 0.10 0.02  100 0 0 0.98 0 1 -30 30"), "taps"),
    (("0.01 0.10 0.02  100 0 0 0 0 1 -30 30", "0.01 0.10 0.02  100 0 0 0 5.0 1 -30 30"), "phase shifters"),
    (("20  1  40.0  10.0  0 0", "20  1  40.0  10.0  0 4.5"), "shunts"),
    (("2 0 0 3  0.02 12.0 100.0", "1 0 0 3  0.02 12.0 100.0"), "piecewise"),
    (("mpc.bus_name", "mpc.dcline = [10 20 1 10 10 5 0 1 1 10 90 -90 90 -90 90 0 0 0 0 0 0 0 0];\nmpc.bus_name"), "HVDC"),
])
def test_unsupported_features_are_rejected_loudly(edit, needle):
    txt = HAND.replace(*edit)
    assert txt != HAND
    with pytest.raises(MP.UnsupportedCase, match=needle):
        MP.network_from_matpower(MP.read_matpower(txt))
    if needle == "HVDC":
        assert MP.network_from_matpower(MP.read_matpower(txt), dcline="drop").nl == 4


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
    with pytest.raises(MP.UnsupportedCase, match="HVDC"):
        MP.network_from_matpower(m)
    net = MP.network_from_matpower(m, dcline="drop")
    assert (net.nb, net.ng, net.nl) == (3, 3, 3) and net.ref_bus == 0     # no type-3 bus in the file: largest generator's bus
    assert np.allclose(net.pd, [1.1, 1.1, 0.95]) and np.allclose(net.rate_a, [90.0, 0.5, 90.0])
    assert np.allclose(net.c2, [1100.0, 850.0, 0.0]) and np.allclose(net.c1, [500.0, 120.0, 0.0])
