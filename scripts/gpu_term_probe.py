"""Termination leg on its own (512 IEEE-118 scenarios, textbook sign, 60 outer iterations at most): wall time, sweeps, work.
Runs in any checkout of the repository (round-3 tree in alt/r03 included): python scripts/gpu_term_probe.py [corrector]"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import torch
import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, contingency, CASES
corr = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nb, ng, nl, seed = CASES["case118"]
base = acopf_synth(nb, ng, nl, seed); lay0 = acopf_layout(base)
opts = pkg.default_options(max_iter=60, literal_quirks=0, ipm_corrector=corr, tol_infeas=1e-6, tol_residual=1e-4, use_soc=1)
ctx = pkg.Context(lay0.n, lay0.m, lay0.num_linear, lay0.jrow, lay0.jcol, lay0.hrow, lay0.hcol, lay0.xL, lay0.xU, lay0.gL, lay0.gU, opts, batch=512)
ctx.acopf_attach(base, lay0)
for b in range(512):
    net = base if b == 0 else contingency(base, b, seed)
    ctx.acopf_set_instance(b, net, acopf_layout(net))
ctx.sqp_reset()
torch.cuda.synchronize()
marks = []
t0 = time.perf_counter()
for k in range(12):
    ctx.sqp_run(5)
    torch.cuda.synchronize()
    c = ctx.counters(); ret, it, done = ctx.sqp_status()
    marks.append((round(time.perf_counter() - t0, 3), int(c["n_sweeps"]), int(c["n_qp"]), int(done.sum())))
print("corrector", corr, "| (seconds, sweeps, qp, done) after every 5 outer iterations:")
for m in marks: print("   ", m)
