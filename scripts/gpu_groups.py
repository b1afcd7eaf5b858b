"""Instance groups on independent contexts / streams, driven by host threads: does overlapping the latency-bound level
launches of several groups raise the throughput?  usage: gpu_groups.py CASE TOTAL STEPS G [G ...]"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, contingency, CASES
case, total, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
nb, ng, nl, seed = CASES[case]
base = acopf_synth(nb, ng, nl, seed); lay0 = acopf_layout(base)
scen = [(base if s == 0 else contingency(base, s, seed)) for s in range(total)]
lays = [acopf_layout(n) for n in scen]
for G in [int(a) for a in sys.argv[4:]]:
    ctxs = []
    for g in range(G):
        lo, hi = g * total // G, (g + 1) * total // G
        ctx = pkg.Context(lay0.n, lay0.m, lay0.num_linear, lay0.jrow, lay0.jcol, lay0.hrow, lay0.hcol, lay0.xL, lay0.xU,
                          lay0.gL, lay0.gU, pkg.default_options(max_iter=3000, tol_infeas=1e-6, tol_residual=1e-4, use_soc=1,
                                                                literal_quirks=int(os.environ.get("LQ", "1"))), batch=hi - lo)
        ctx.acopf_attach(base, lay0)
        for b, s in enumerate(range(lo, hi)):
            ctx.acopf_set_instance(b, scen[s], lays[s])
        ctx.sqp_reset()
        ctxs.append(ctx)
    def run(k):
        ths = [threading.Thread(target=c.sqp_run, args=(k,)) for c in ctxs]
        [t.start() for t in ths]; [t.join() for t in ths]
    run(1)
    q0 = sum(c.counters()["n_qp"] for c in ctxs)
    t0 = time.time(); run(steps); dt = time.time() - t0
    q1 = sum(c.counters()["n_qp"] for c in ctxs)
    print(f"{case} total {total} groups {G}: {dt*1e3:.0f} ms for {steps} steps, {q1-q0} QPs => {(q1-q0)/dt:.0f} QP/s", flush=True)
    [c.close() for c in ctxs]
