"""Straggler analysis of a batched run: per outer step, the largest interior-point iteration count over the instances
(= the sweeps the step needs) against the mean.  usage: gpu_stragglers.py CASE BATCH STEPS [key=value ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, contingency, CASES
case, B, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
kw = dict(max_iter=3000, tol_infeas=1e-6, tol_residual=1e-4, use_soc=1)
for a in sys.argv[4:]:
    k, v = a.split("="); kw[k] = float(v) if "." in v or "e" in v else int(v)
nb, ng, nl, seed = CASES[case]
base = acopf_synth(nb, ng, nl, seed); lay0 = acopf_layout(base)
ctx = pkg.Context(lay0.n, lay0.m, lay0.num_linear, lay0.jrow, lay0.jcol, lay0.hrow, lay0.hcol, lay0.xL, lay0.xU, lay0.gL,
                  lay0.gU, pkg.default_options(**kw), batch=B)
ctx.acopf_attach(base, lay0)
for b in range(B):
    net = base if b == 0 else contingency(base, b, seed)
    ctx.acopf_set_instance(b, net, acopf_layout(net))
ctx.sqp_reset()
t0 = time.time(); ctx.sqp_run(steps); dt = time.time() - t0
c = ctx.counters()
tr = [ctx.sqp_trace(b) for b in range(B)]
it = np.zeros((B, steps + 2))
for b in range(B):
    for r in tr[b]:
        if r["iter"] <= steps: it[b, r["iter"]] += r["ipm_iters"]
print(f"{case} B={B} {steps} steps: {dt*1e3:.0f} ms, n_qp {c['n_qp']} ipm {c['n_ipm_iter']} fac {c['n_factor']} => {c['n_qp']/dt:.0f} QP/s")
for k in range(1, steps + 1):
    col = it[:, k]
    top = np.argsort(-col)[:4]
    print(f"  step {k}: mean ipm its {col.mean():6.1f}  p90 {np.percentile(col, 90):5.0f}  max {col.max():5.0f}  top instances {[(int(b), int(col[b])) for b in top]}")
print("  sum of per-step maxima", it[:, 1:steps+1].max(axis=0).sum(), " sum of means", it[:, 1:steps+1].mean(axis=0).sum())
ctx.close()
