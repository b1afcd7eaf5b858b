"""Runs the eight 64-scenario shards of the 512-scenario IEEE-118 set one after the other on ONE GPU (what ranks
0..7 of `bench.py --gpus 8` each get): K outer iterations per shard, work counters, throughput, return codes and
sub-problem statuses -- a robustness check of the scenarios the single-GPU bench never touches."""
import sys, os, time, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, contingency, CASES
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
quirks = int(sys.argv[2]) if len(sys.argv) > 2 else 1
nb, ng, nl, seed = CASES["case118"]
base = acopf_synth(nb, ng, nl, seed); lay0 = acopf_layout(base)
opts = pkg.default_options(tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, max_iter=3000, literal_quirks=quirks)
for shard in range(8):
    ctx = pkg.Context(lay0.n, lay0.m, lay0.num_linear, lay0.jrow, lay0.jcol, lay0.hrow, lay0.hcol, lay0.xL, lay0.xU,
                      lay0.gL, lay0.gU, opts, batch=64)
    ctx.acopf_attach(base, lay0)
    for b in range(64):
        s = 64 * shard + b
        net = base if s == 0 else contingency(base, s, seed)
        ctx.acopf_set_instance(b, net, acopf_layout(net))
    ctx.sqp_reset(); ctx.sqp_run(1)
    c0 = ctx.counters(); t = time.time(); ctx.sqp_run(K); t = time.time() - t; c1 = ctx.counters()
    ret, it, done = ctx.sqp_status()
    sub = collections.Counter()
    finite = True
    for b in range(64):
        for r in ctx.sqp_trace(b):
            sub[r["sub_status"]] += 1
        finite &= bool(np.isfinite(ctx.sqp_get(b)["x"]).all())
    nq = c1["n_qp"] - c0["n_qp"]
    print(f"shard {shard}: {nq} QPs in {t:.2f}s = {nq / t:.1f} QP/s, ipm/qp {(c1['n_ipm_iter'] - c0['n_ipm_iter']) / nq:.1f}, "
          f"fac {c1['n_factor'] - c0['n_factor']}, ret {dict(collections.Counter(ret.tolist()))}, iters {it.min()}..{it.max()}, "
          f"sub-problem statuses {dict(sub)}, finite {finite}", flush=True)
    ctx.close()
