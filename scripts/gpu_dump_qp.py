"""Run one scenario of the bench workload alone for K outer iterations and dump the sub-problem it worked on last
(gpurun_out/qp_<inst>_<K>.npz) together with its per-sub-problem log.  usage: gpu_dump_qp.py INST K"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, contingency, CASES
inst, K = int(sys.argv[1]), int(sys.argv[2])
nb, ng, nl, seed = CASES["case118"]
base = acopf_synth(nb, ng, nl, seed); lay0 = acopf_layout(base)
net = base if inst == 0 else contingency(base, inst, seed); lay = acopf_layout(net)
kw = dict(max_iter=3000, tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, literal_quirks=1)
ctx = pkg.Context(lay0.n, lay0.m, lay0.num_linear, lay0.jrow, lay0.jcol, lay0.hrow, lay0.hcol, lay0.xL, lay0.xU, lay0.gL,
                  lay0.gU, pkg.default_options(**kw), batch=1)
ctx.acopf_attach(base, lay0); ctx.acopf_set_instance(0, net, lay)
ctx.sqp_reset(); ctx.sqp_run(1); ctx.sqp_run(K - 1)
log = ctx.sqp_qp_log(0)
print("sub-problems (mode, status, ipm, fac):", log)
rq = ctx.sqp_last_request(0)
print("last request: mode", rq["mode"], "delta", rq["delta"], "mu_pen", rq["mu_pen"])
os.makedirs("gpurun_out", exist_ok=True)
np.savez(f"gpurun_out/qp_{inst}_{K}.npz", log=np.array(log), **rq)
ctx.close()
