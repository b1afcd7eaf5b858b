"""The geographic generator of round 3 (sqpsolver.jl_amd/acopf_synth.py, acopf_synth_geo): structure of what it builds
and -- through the CPU oracle -- that the NLPs it gives are AC-feasible and converge from the flat start, which the
SURVEY.md section 8d recipe does not achieve at 1354 / 9241 buses."""
import numpy as np

from sqpsolver_jl_amd.acopf_synth import acopf_synth_geo, acopf_layout, contingency, synth_case, CASES
from oracle import oracle as O


def _connected(nb, f, t):
    adj = [[] for _ in range(nb)]
    for a, b in zip(f, t):
        adj[a].append(b); adj[b].append(a)
    seen = np.zeros(nb, bool); stack = [0]; seen[0] = True
    while stack:
        u = stack.pop()
        for v in adj[u]:
            if not seen[v]:
                seen[v] = True; stack.append(v)
    return bool(seen.all())


def test_geo_network_structure():
    nb, ng, nl, seed = CASES["case1354"]
    a, b = acopf_synth_geo(nb, ng, nl, seed), acopf_synth_geo(nb, ng, nl, seed)
    assert (a.nb, a.ng, a.nl) == (nb, ng, nl) and len(a.f_bus) == nl and len(a.gen_bus) == ng
    for k in ("f_bus", "t_bus", "r", "x", "bc", "pd", "qd", "pmax", "rate_a", "gen_bus"):
        assert np.array_equal(getattr(a, k), getattr(b, k)), k                 # deterministic per seed
    assert _connected(nb, a.f_bus, a.t_bus)
    assert len(set(zip(a.f_bus.tolist(), a.t_bus.tolist()))) == nl               # no parallel branches
    # every branch is a lattice edge of the 5-wide strip: neighbours in a column or in a row
    d = a.t_bus.astype(int) - a.f_bus.astype(int)
    assert set(np.unique(d).tolist()) <= {1, 5}
    assert np.all(a.rate_a > 0) and np.all(a.x >= 3 * a.r - 1e-15) and np.all(a.pmax > 0)
    # generation is local: every stretch of nb / ng buses holds one generator
    cuts = np.linspace(0, nb, ng + 1).astype(int)
    assert all(cuts[g] <= a.gen_bus[g] < max(cuts[g] + 1, cuts[g + 1]) for g in range(ng))
    # the case table picks it for the large shapes only
    assert synth_case("case118").nl == 186 and np.array_equal(synth_case("case1354").f_bus, a.f_bus)
    lay = acopf_layout(a)
    assert lay.n == 2 * nb + 2 * ng + 4 * nl and lay.m == 1 + 2 * nb + 8 * nl


def test_geo_network_converges_from_the_flat_start():
    """300 buses, 60 generators, 450 branches: the oracle's SQP-TR converges (textbook Hessian sign) in a handful of
    iterations, for the base case and a contingency; the restoration phase is left after the first iterations."""
    base = acopf_synth_geo(300, 60, 450, 7)
    for net in (base, contingency(base, 11, 7)):
        lay = acopf_layout(net)
        r = O.sqp_solve(O.problem_acopf(net, lay), O.default_options(kkt_mode=2, max_iter=40, tol_infeas=1e-6, tol_residual=1e-4,
                                                                    use_soc=1, literal_quirks=0, num_threads=4))
        assert r["status"] == 0 and r["iter"] <= 25, (r["status"], r["iter"])
        g = O.problem_acopf(net, lay).eval_g(r["x"])
        viol = np.maximum(0.0, np.maximum(lay.gL - g, g - lay.gU)).sum() + np.maximum(0.0, np.maximum(lay.xL - r["x"], r["x"] - lay.xU)).sum()
        assert viol <= 1e-6
