"""Device vs oracle on the bench workload: per-sub-problem interior-point iteration counts of the first outer iterations."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sqpsolver_jl_amd as pkg
from oracle import oracle as O
from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, contingency, CASES
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3
nb, ng, nl, seed = CASES["case118"]
base = acopf_synth(nb, ng, nl, seed)
nets = [base, contingency(base, 7, seed)]
lays = [acopf_layout(nt) for nt in nets]
kw = dict(max_iter=iters, tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, literal_quirks=int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = pkg.Context(lays[0].n, lays[0].m, lays[0].num_linear, lays[0].jrow, lays[0].jcol, lays[0].hrow, lays[0].hcol,
                  lays[0].xL, lays[0].xU, lays[0].gL, lays[0].gU, pkg.default_options(**kw), batch=2)
ctx.acopf_attach(base, lays[0])
for b in range(2):
    ctx.acopf_set_instance(b, nets[b], lays[b])
ctx.sqp_reset(); ctx.sqp_run(0)
for b in range(2):
    ro = O.sqp_solve(O.problem_acopf(nets[b], lays[b]), O.default_options(num_threads=16, **kw))
    rg = ctx.sqp_get(b); tr = ctx.sqp_trace(b)
    print("inst", b, "oracle", [(a["iter"], a["sub_status"], a["ipm_iters"]) for a in ro["trace"]])
    print("        device", [(t["iter"], t["sub_status"], t["ipm_iters"]) for t in tr])
    print("        rel dx", np.abs(rg["x"] - ro["x"]).max() / np.abs(ro["x"]).max(), "obj", rg["obj_val"], ro["obj_val"])
