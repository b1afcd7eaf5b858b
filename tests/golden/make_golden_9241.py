"""Golden fixture of BASELINE.json configs[4] as bench.py runs it: scenario 1 (a line outage) of the geographic 9241-bus shape
through the CPU oracle -- (a) the reference's Hessian sign (literal_quirks = 1, the bench default) over the bench's first
W + K = 7 outer iterations, (b) the textbook sign to convergence.  Per run: status, outer iterations, objective, the decision
sequence and radii of the trace, the final point, and one row per sub-problem (mode, MOI status, interior-point iterations,
factorisations, rule that ended it: 0 tolerance, 1 / 2 / 3 acceptable-termination rules, final scaled error).  Oracle outputs,
NOT reference outputs (the reference cannot run here).  Takes ~10 minutes (one core):  python tests/golden/make_golden_9241.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import sqpsolver_jl_amd  # noqa: E402,F401
from sqpsolver_jl_amd.acopf_synth import synth_case, acopf_layout, contingency, CASES  # noqa: E402
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SCEN = 1


def run(net, lay, **kw):
    log = os.path.join("/tmp", f"qplog_9241_{os.getpid()}.txt")
    if os.path.exists(log):
        os.remove(log)
    os.environ["ORA_QP_LOG"] = log
    t0 = time.time()
    r = O.sqp_solve(O.problem_acopf(net, lay), O.default_options(tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, kkt_mode=2,
                                                                   num_threads=1, **kw))
    del os.environ["ORA_QP_LOG"]
    qps = np.loadtxt(log, ndmin=2)
    os.remove(log)
    print(f"  {kw}: status {r['status']} iter {r['iter']} n_qp {r['n_qp']} obj {r['obj_val']:.10e}  ({time.time() - t0:.0f} s)")
    return dict(status=np.int64(r["status"]), iter=np.int64(r["iter"]), obj_val=np.float64(r["obj_val"]), x=r["x"],
                trace=np.array([[t["iter"], t["accepted"], t["fr"], t["sub_status"], t["delta"], t["pnorm"], t["prim_infeas"]] for t in r["trace"]]),
                # columns: mode, MOI status, ipm iterations, factorisations, outer iteration, radius, rule, final scaled error
                qps=qps[:, [0, 1, 2, 3, 4, 5, 7, 8]])


def main():
    nb, ng, nl, seed = CASES["case9241"]
    base = synth_case("case9241", "geo")
    net = contingency(base, SCEN, seed)
    lay = acopf_layout(net)
    out = {}
    for tag, kw in (("quirks1_7it", dict(literal_quirks=1, max_iter=7)), ("quirks0_conv", dict(literal_quirks=0, max_iter=60))):
        for k, v in run(net, lay, **kw).items():
            out[f"{tag}_{k}"] = v
    np.savez_compressed(os.path.join(HERE, "geo9241_s1.npz"), **out)
    print("wrote geo9241_s1.npz", {k: np.shape(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
