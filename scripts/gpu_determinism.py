"""Run the same batched SQP twice in one process and compare the iterates bit for bit:
python scripts/gpu_determinism.py case B iters"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, contingency, CASES
case = sys.argv[1]; B = int(sys.argv[2]); iters = int(sys.argv[3])
nb, ng, nl, seed = CASES[case]
base = acopf_synth(nb, ng, nl, seed); lay0 = acopf_layout(base)
opts = pkg.default_options(tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, max_iter=iters, literal_quirks=1)
trs = []
def run():
    ctx = pkg.Context(lay0.n, lay0.m, lay0.num_linear, lay0.jrow, lay0.jcol, lay0.hrow, lay0.hcol, lay0.xL, lay0.xU, lay0.gL, lay0.gU, opts, batch=B)
    ctx.acopf_attach(base, lay0)
    for b in range(B):
        net = base if b == 0 else contingency(base, b, seed)
        ctx.acopf_set_instance(b, net, acopf_layout(net))
    ctx.sqp_reset(); ctx.sqp_run(0)
    xs = [ctx.sqp_get(b)["x"].copy() for b in range(B)]
    trs.append([[(r["iter"], r["sub_status"], r["fr"], r["ipm_iters"], r["accepted"]) for r in ctx.sqp_trace(b)] for b in range(B)])
    c = ctx.counters(); ctx.close()
    return xs, (c["n_qp"], c["n_ipm_iter"], c["n_factor"])
a, ca = run(); b, cb = run()
diff = [float(np.max(np.abs(x - y))) for x, y in zip(a, b)]
print("counters", ca, cb, "max |dx| per instance:", ["%.1e" % d for d in diff])
for b in range(B):
    if trs[0][b] != trs[1][b]:
        print(" instance", b, "run1", trs[0][b], "run2", trs[1][b])
print("DETERMINISTIC" if max(diff) == 0.0 and ca == cb else "NOT deterministic")
