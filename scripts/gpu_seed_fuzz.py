"""Robustness: other synthetic networks than the ones the tests pin (seeds), device against oracle to convergence.
usage: gpu_seed_fuzz.py CASE NSEEDS"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, acr_layout, contingency, CASES
from oracle import oracle as O
case, ns = sys.argv[1], int(sys.argv[2])
nb, ng, nl, seed0 = CASES[case]
kw = dict(max_iter=60, tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, literal_quirks=0)
bad = 0
for seed in range(seed0 + 1, seed0 + 1 + ns):
    for form, layout in (("polar", acopf_layout), ("acr", acr_layout)):
        base = acopf_synth(nb, ng, nl, seed)
        nets = [base, contingency(base, 1, seed), contingency(base, 2, seed)]
        lays = [layout(nt) for nt in nets]
        ctx = pkg.Context(lays[0].n, lays[0].m, lays[0].num_linear, lays[0].jrow, lays[0].jcol, lays[0].hrow, lays[0].hcol,
                          lays[0].xL, lays[0].xU, lays[0].gL, lays[0].gU, pkg.default_options(**kw), batch=3)
        ctx.acopf_attach(base, lays[0])
        for b in range(3):
            ctx.acopf_set_instance(b, nets[b], lays[b])
        ctx.sqp_reset(); ctx.sqp_run(0)
        for b in range(3):
            rg = ctx.sqp_get(b)
            ro = O.sqp_solve(O.problem_acopf(nets[b], lays[b]), O.default_options(num_threads=4, **kw))
            same = (rg["status"], rg["iter"]) == (ro["status"], ro["iter"])
            dx = np.abs(rg["x"] - ro["x"]).max() / max(1.0, np.abs(ro["x"]).max())
            ok = same and (dx < 1e-6 or ro["status"] != 0)
            bad += not ok
            print(f"seed {seed} {form} inst {b}: device ({rg['status']}, {rg['iter']}) oracle ({ro['status']}, {ro['iter']}) |dx| {dx:.1e} {'ok' if ok else 'DIFF'}", flush=True)
        ctx.close()
print("differences:", bad)
