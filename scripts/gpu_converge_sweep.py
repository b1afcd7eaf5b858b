"""All 512 scenarios of the IEEE-118 set to termination (textbook Hessian sign), shard by shard on one GPU:
return codes, iteration statistics, throughput."""
import sys, os, time, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, contingency, CASES
quirks = int(sys.argv[1]) if len(sys.argv) > 1 else 0
max_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 60
nb, ng, nl, seed = CASES["case118"]
base = acopf_synth(nb, ng, nl, seed); lay0 = acopf_layout(base)
opts = pkg.default_options(tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, max_iter=max_iter, literal_quirks=quirks)
tot = collections.Counter(); nq = 0; tt = 0.0
for shard in range(8):
    ctx = pkg.Context(lay0.n, lay0.m, lay0.num_linear, lay0.jrow, lay0.jcol, lay0.hrow, lay0.hcol, lay0.xL, lay0.xU,
                      lay0.gL, lay0.gU, opts, batch=64)
    ctx.acopf_attach(base, lay0)
    for b in range(64):
        s = 64 * shard + b
        net = base if s == 0 else contingency(base, s, seed)
        ctx.acopf_set_instance(b, net, acopf_layout(net))
    ctx.sqp_reset(); t = time.time(); ctx.sqp_run(0); t = time.time() - t
    ret, it, done = ctx.sqp_status(); c = ctx.counters()
    finite = all(np.isfinite(ctx.sqp_get(b)["x"]).all() for b in range(64))
    tot.update(ret.tolist()); nq += c["n_qp"]; tt += t
    print(f"shard {shard}: {t:.1f}s ret {dict(collections.Counter(ret.tolist()))} iters {it.min()}/{it.mean():.1f}/{it.max()} "
          f"n_qp {c['n_qp']} ipm/qp {c['n_ipm_iter'] / c['n_qp']:.1f} QP/s {c['n_qp'] / t:.0f} all done {bool(done.all())} finite {finite}", flush=True)
    ctx.close()
print(f"total: {dict(tot)}  {nq} sub-problems in {tt:.1f}s = {nq / tt:.0f} QP/s")
