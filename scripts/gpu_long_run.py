"""Long batched run: 64 x IEEE-118 scenarios for many outer iterations; reports the return codes."""
import sys, os, time, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, contingency, CASES
case = sys.argv[1]; B = int(sys.argv[2]); iters = int(sys.argv[3]); quirks = int(sys.argv[4])
extra = {"ipm_corrector": int(sys.argv[5])} if len(sys.argv) > 5 else {}
nb, ng, nl, seed = CASES[case]
base = acopf_synth(nb, ng, nl, seed); lay0 = acopf_layout(base)
opts = pkg.default_options(tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, max_iter=iters, literal_quirks=quirks, **extra)
ctx = pkg.Context(lay0.n, lay0.m, lay0.num_linear, lay0.jrow, lay0.jcol, lay0.hrow, lay0.hcol, lay0.xL, lay0.xU, lay0.gL, lay0.gU, opts, batch=B)
ctx.acopf_attach(base, lay0)
for b in range(B):
    net = base if b == 0 else contingency(base, b, seed)
    ctx.acopf_set_instance(b, net, acopf_layout(net))
ctx.sqp_reset(); t0 = time.time(); ctx.sqp_run(0); t = time.time() - t0
ret, it, done = ctx.sqp_status()
c = ctx.counters()
print(f"{case} B={B} max_iter={iters} quirks={quirks} {extra}: {t:.1f}s  ret codes {dict(collections.Counter(ret.tolist()))}  iters min/mean/max {it.min()}/{it.mean():.1f}/{it.max()}  n_qp {c['n_qp']} ipm/qp {c['n_ipm_iter']/c['n_qp']:.1f} QP/s {c['n_qp']/t:.1f}")
bad = [b for b in range(B) if ret[b] == -5]
for b in bad[:3]:
    tr = ctx.sqp_trace(b)
    print("  inst", b, "last rows:", [(r["iter"], r["sub_status"], r["fr"], round(r["delta"], 6), r["ipm_iters"]) for r in tr[-3:]])
obj = [ctx.sqp_get(b)["obj_val"] for b in range(min(B, 4))]
print("  obj", obj)
