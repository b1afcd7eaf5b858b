"""IPM iteration counts per sub-problem, device (sparse and dense solver) vs oracle (dense natural order and its own
sparse order): data behind the tolerance of the IEEE-118 parity tests."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, contingency, CASES
from oracle import oracle as O
nb, ng, nl, seed = CASES["case118"]
base = acopf_synth(nb, ng, nl, seed)
mi = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for lq in (1, 0):
    nets = [base, contingency(base, 7, seed), contingency(base, 3, seed)]
    lays = [acopf_layout(nt) for nt in nets]
    kw = dict(max_iter=mi, tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, literal_quirks=lq)
    dev = {}
    for mode in (2, 1):
        ctx = pkg.Context(lays[0].n, lays[0].m, lays[0].num_linear, lays[0].jrow, lays[0].jcol, lays[0].hrow, lays[0].hcol,
                          lays[0].xL, lays[0].xU, lays[0].gL, lays[0].gU, pkg.default_options(kkt_mode=mode, **kw), batch=len(nets))
        ctx.acopf_attach(base, lays[0])
        for b in range(len(nets)):
            ctx.acopf_set_instance(b, nets[b], lays[b])
        ctx.sqp_reset(); ctx.sqp_run(0)
        dev[mode] = [([t["ipm_iters"] for t in ctx.sqp_trace(b)], ctx.sqp_get(b)["x"]) for b in range(len(nets))]
        ctx.close()
    for b in range(len(nets)):
        ora = {m: O.sqp_solve(O.problem_acopf(nets[b], lays[b]), O.default_options(kkt_mode=m, num_threads=16, **kw)) for m in (2, 1)}
        print(f"lq={lq} inst {b}:")
        print("   device sparse ", dev[2][b][0])
        print("   device dense  ", dev[1][b][0])
        print("   oracle sparse ", [t["ipm_iters"] for t in ora[2]["trace"]])
        print("   oracle dense  ", [t["ipm_iters"] for t in ora[1]["trace"]])
        print("   |x dev_sparse - ora_sparse| %.2e  |x dev_dense - ora_dense| %.2e  |x ora_sparse - ora_dense| %.2e" % (
            np.abs(dev[2][b][1] - ora[2]["x"]).max(), np.abs(dev[1][b][1] - ora[1]["x"]).max(), np.abs(ora[2]["x"] - ora[1]["x"]).max()), flush=True)
