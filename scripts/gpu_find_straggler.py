"""Find the sub-problem that keeps one instance busy long after the rest of the batch (bench workload, first B scenarios).
usage: gpu_find_straggler.py B STEPS [dump_dir]"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, contingency, CASES
B, steps = int(sys.argv[1]), int(sys.argv[2])
nb, ng, nl, seed = CASES["case118"]
base = acopf_synth(nb, ng, nl, seed); lay0 = acopf_layout(base)
kw = dict(max_iter=3000, tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, literal_quirks=1)
ctx = pkg.Context(lay0.n, lay0.m, lay0.num_linear, lay0.jrow, lay0.jcol, lay0.hrow, lay0.hcol, lay0.xL, lay0.xU, lay0.gL,
                  lay0.gU, pkg.default_options(**kw), batch=B)
ctx.acopf_attach(base, lay0)
for b in range(B):
    net = base if b == 0 else contingency(base, b, seed)
    ctx.acopf_set_instance(b, net, acopf_layout(net))
ctx.sqp_reset(); ctx.sqp_run(1); ctx.sqp_run(steps - 1)
qp, ipm, fac = ctx.sqp_work()
order = np.argsort(-fac)
print("factorisations per instance: mean %.0f p90 %.0f max %d; ipm mean %.0f max %d" % (fac.mean(), np.percentile(fac, 90), fac.max(), ipm.mean(), ipm.max()))
print("busiest instances (inst, qp, ipm, fac):", [(int(b), int(qp[b]), int(ipm[b]), int(fac[b])) for b in order[:6]])
worst = []
for b in range(B):
    tr = ctx.sqp_trace(b)
    for r in tr:
        worst.append((r["ipm_iters"], b, r["iter"], r["sub_status"], r["fr"], r["delta"]))
worst.sort(reverse=True)
print("largest (ipm_iters, instance, outer iter, sub_status, fr, delta):")
for w in worst[:8]:
    print("   ", w)
b = int(order[0])
print(f"sub-problems of instance {b} (mode, status, ipm, fac):", ctx.sqp_qp_log(b))
print(f"trace of instance {b}:")
for r in ctx.sqp_trace(b):
    print("   ", {k: (float(f"{v:.4g}") if isinstance(v, float) else v) for k, v in r.items()})
ctx.close()
