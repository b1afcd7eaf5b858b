import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, contingency, CASES
case = sys.argv[1]; B = int(sys.argv[2]); iters = int(sys.argv[3]); quirks = int(sys.argv[4]); scale = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
nb, ng, nl, seed = CASES[case]
base = acopf_synth(nb, ng, nl, seed, load_scale=scale); lay0 = acopf_layout(base)
opts = pkg.default_options(tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, max_iter=iters, literal_quirks=quirks)
ctx = pkg.Context(lay0.n, lay0.m, lay0.num_linear, lay0.jrow, lay0.jcol, lay0.hrow, lay0.hcol, lay0.xL, lay0.xU, lay0.gL, lay0.gU, opts, batch=B)
ctx.acopf_attach(base, lay0)
for b in range(B):
    net = base if b == 0 else contingency(base, b, seed)
    ctx.acopf_set_instance(b, net, acopf_layout(net))
ctx.sqp_reset(); t0 = time.time(); ctx.sqp_run(0); t = time.time() - t0
ret, it, done = ctx.sqp_status(); print("ret", ret.tolist(), "iters", it.tolist(), "time %.1f" % t)
for r in ctx.sqp_trace(0):
    print("  it%3d %s%s st%2d ipm%3d f=%.6e phi=%.4e mu=%.2e D=%.3e |p|=%.3e pr=%.3e du=%.2e" % (r["iter"], "a" if r["accepted"] else "r", "F" if r["fr"] else " ", r["sub_status"], r["ipm_iters"], r["f"], r["phi"], r["mu"], r["delta"], r["pnorm"], r["prim_infeas"], r["dual_infeas"]))
