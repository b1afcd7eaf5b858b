"""Regenerates tests/golden/*.json.

The reference (Julia + JuMP + Ipopt) cannot be executed in this image, so these fixtures are of two
kinds, kept apart in the files:
  "reference_pins": known answers read off the reference's own tests / README
                    (/root/reference/test/runtests.jl:12-14, README.md:18-21) and hand-derived KKT facts
                    (SURVEY.md section 8c, KAT-1..KAT-5);
  "oracle_runs":    outputs of the CPU oracle (oracle/), pinned so that later changes to the oracle or
                    to the HIP path are caught.  They are NOT reference outputs.
Run:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sqpsolver_jl_amd  # noqa: E402,F401
from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, contingency, CASES  # noqa: E402
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def run(prob, **kw):
    r = O.sqp_solve(prob, O.default_options(**kw))
    return dict(status=r["status"], iter=r["iter"], obj_val=r["obj_val"], x=r["x"].tolist(),
                mult_g=r["mult_g"].tolist(), n_qp=r["n_qp"],
                trace=[[t["iter"], t["accepted"], t["fr"], t["sub_status"], t["delta"], t["pnorm"],
                        t["prim_infeas"], t["mu"]] for t in r["trace"]])


def main():
    pins = {
        "toy": {"source": "/root/reference/test/ext_solver.jl:14-28, test/runtests.jl:12-14",
                "x": [-1.0, -1.0], "rtol": 1e-4, "f": 0.0, "n": 2, "m": 4, "num_linear": 1,
                "g_L": [-2.0, 0.0, 0.0, 0.0], "g_U": ["inf", 0.0, 0.0, "inf"],
                "lambda_jump_at_solution": [0.0, 1.0 / 3.0, 0.0, 0.0],
                "first_iterate": {"E": [0.0, -2.0, -1.0, 0.0], "viol1": 3.0, "qp_status": "LOCALLY_INFEASIBLE",
                                  "fr_lp_optimum": 1.0, "fr_p1": -2.0}},
        "readme1": {"source": "/root/reference/README.md:18-21,37-39", "x": [-1.0], "rtol": 1e-4},
        "hs071": {"source": "Hock-Schittkowski 71 (MathOptInterface nonlinear tests, not vendored)",
                  "x": [1.0, 4.7429996, 3.8211500, 1.3794083], "f": 17.0140173, "rtol": 1e-4},
    }
    runs = {
        "toy": run(O.problem_toy(), max_iter=100),
        "readme1": run(O.problem_readme1(), max_iter=100),
        "hs071": run(O.problem_hs071(), max_iter=200),
    }
    nb, ng, nl, seed = CASES["case14"]
    base = acopf_synth(nb, ng, nl, seed)
    for tag, net in (("case14_s0", base), ("case14_s3", contingency(base, 3, seed))):
        lay = acopf_layout(net)
        for q in (1, 0):
            runs[f"{tag}_quirks{q}"] = run(O.problem_acopf(net, lay), max_iter=25, tol_infeas=1e-6,
                                          tol_residual=1e-4, literal_quirks=q)
    json.dump({"reference_pins": pins}, open(os.path.join(HERE, "reference_pins.json"), "w"), indent=1)
    json.dump({"oracle_runs": runs}, open(os.path.join(HERE, "oracle_runs.json"), "w"))
    # network fingerprint: the synthetic generator must be bit-stable across machines
    fp = {}
    for name, (nb, ng, nl, seed) in CASES.items():
        if nb > 2000:
            continue
        net = acopf_synth(nb, ng, nl, seed)
        lay = acopf_layout(net)
        fp[name] = dict(n=lay.n, m=lay.m, nnzj=len(lay.jrow), nnzh=len(lay.hrow),
                        sum_pd=float(net.pd.sum()), sum_rate=float(net.rate_a.sum()),
                        f_bus_head=net.f_bus[:8].tolist(), t_bus_head=net.t_bus[:8].tolist())
    json.dump({"synthetic_networks": fp}, open(os.path.join(HERE, "networks.json"), "w"), indent=1)
    # the 14-bus synthetic case in the reference's on-disk format (MATPOWER v2), read back by tests/test_matpower.py
    from sqpsolver_jl_amd.matpower import write_matpower
    with open(os.path.join(HERE, "case14_synth.m"), "w") as fh:
        fh.write("% synthetic IEEE-14-shaped case written by tests/golden/make_golden.py (acopf_synth(14,5,20,seed=14))\n")
        fh.write(write_matpower(base, "case14_synth"))
    # the reference's example case, re-serialised through our own writer (data, not the reference's file), together
    # with the dispatch the file itself stores in its gen / dcline columns: a reference-derived pin for the full
    # ACOPF + HVDC model (examples/acopf/opf.jl:12-46)
    ref_case = "/root/reference/examples/acopf/case3.m"
    if os.path.exists(ref_case):
        from sqpsolver_jl_amd.matpower import read_matpower, network_from_matpower
        m3 = read_matpower(ref_case)
        net3 = network_from_matpower(m3)
        with open(os.path.join(HERE, "case3_network.m"), "w") as fh:
            fh.write("% network data of the reference's examples/acopf/case3.m, re-serialised by tests/golden/make_golden.py\n")
            fh.write(write_matpower(net3, "case3_network", base_mva=float(m3["baseMVA"])))
        json.dump({"case3_dispatch": {"source": "gen column Pg and dcline columns Pf, Pt of /root/reference/examples/acopf/case3.m",
                                      "pg_mw": m3["gen"][:, 1].tolist(), "dc_pf_mw": float(m3["dcline"][0, 3]),
                                      "dc_pt_mw": float(m3["dcline"][0, 4]), "base_mva": float(m3["baseMVA"]),
                                      "objective": 5906.88, "objective_source": "this build's optimum, consistent with "
                                      "the value PowerModels documents for the case (5907)"}},
                  open(os.path.join(HERE, "case3_dispatch.json"), "w"), indent=1)
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
