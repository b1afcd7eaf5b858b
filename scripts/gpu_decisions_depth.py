"""How long do device and oracle take the same decisions on the bench workload (reference Hessian sign)?
usage: gpu_decisions_depth.py NSCEN ITERS [LQ]  -- random scenarios of the 512-scenario bench set, device run as one batch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, contingency, CASES
from oracle import oracle as O
ns, iters = int(sys.argv[1]), int(sys.argv[2])
lq = int(sys.argv[3]) if len(sys.argv) > 3 else 1
nb, ng, nl, seed = CASES["case118"]
base = acopf_synth(nb, ng, nl, seed)
ids = sorted(np.random.default_rng(3).choice(512, size=ns, replace=False).tolist())
nets = [base if s == 0 else contingency(base, s, seed) for s in ids]
lays = [acopf_layout(nt) for nt in nets]
kw = dict(max_iter=iters, tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, literal_quirks=lq)
ctx = pkg.Context(lays[0].n, lays[0].m, lays[0].num_linear, lays[0].jrow, lays[0].jcol, lays[0].hrow, lays[0].hcol,
                  lays[0].xL, lays[0].xU, lays[0].gL, lays[0].gU, pkg.default_options(**kw), batch=ns)
ctx.acopf_attach(base, lays[0])
for b in range(ns):
    ctx.acopf_set_instance(b, nets[b], lays[b])
ctx.sqp_reset(); ctx.sqp_run(0)
for b in range(ns):
    ro = O.sqp_solve(O.problem_acopf(nets[b], lays[b]), O.default_options(kkt_mode=2, num_threads=1, **kw))
    tr = ctx.sqp_trace(b); rg = ctx.sqp_get(b)
    A = [(a["iter"], a["accepted"], a["fr"], a["sub_status"]) for a in ro["trace"]]
    T = [(t["iter"], t["accepted"], t["fr"], t["sub_status"]) for t in tr]
    first = next((k for k in range(min(len(A), len(T))) if A[k] != T[k]), None)
    first_fr = next((k for k, a in enumerate(A) if a[2]), None)
    ipm_o = [a["ipm_iters"] for a in ro["trace"]]; ipm_d = [t["ipm_iters"] for t in tr]
    dx = np.abs(rg["x"] - ro["x"]).max() / max(1.0, np.abs(ro["x"]).max())
    print(f"scen {ids[b]}: rows {len(A)}/{len(T)} first differing decision {first} first FR row {first_fr} status {ro['status']}/{rg['status']} "
          f"iter {ro['iter']}/{rg['iter']} |dx| {dx:.1e} ipm diff max {max(abs(a - t) for a, t in zip(ipm_o, ipm_d))}", flush=True)
