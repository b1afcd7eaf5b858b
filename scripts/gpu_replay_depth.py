"""Every sub-problem the batched device run solves for a few random scenarios of the 512-scenario bench set, replayed
through the drop-in seat of a fresh one-instance context and through the oracle's seat: data for the parity-depth test.
usage: gpu_replay_depth.py NSCEN ITERS BATCH [LQ]"""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, contingency, CASES
from oracle import oracle as O
import test_gpu_parity as T
ns, iters, batch = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
lq = int(sys.argv[4]) if len(sys.argv) > 4 else 1
nb, ng, nl, seed = CASES["case118"]
base = acopf_synth(nb, ng, nl, seed)
lay0 = acopf_layout(base)
kw = dict(max_iter=3000, tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, literal_quirks=lq)
ctx = pkg.Context(lay0.n, lay0.m, lay0.num_linear, lay0.jrow, lay0.jcol, lay0.hrow, lay0.hcol, lay0.xL, lay0.xU, lay0.gL, lay0.gU,
                  pkg.default_options(**kw), batch=batch)
ctx.acopf_attach(base, lay0)
scen = {}
t0 = time.time()
for b in range(batch):
    net = base if b == 0 else contingency(base, b, seed)
    scen[b] = (net, acopf_layout(net))
    ctx.acopf_set_instance(b, *scen[b])
print(f"set-up {time.time() - t0:.0f} s", flush=True)
ids = sorted(np.random.default_rng(5).choice(batch, size=ns, replace=False).tolist())
seats, oseats = {}, {}
for b in ids:
    lay = scen[b][1]
    seats[b] = pkg.Context(lay.n, lay.m, lay.num_linear, lay.jrow, lay.jcol, lay.hrow, lay.hcol, lay.xL, lay.xU, lay.gL, lay.gU,
                           pkg.default_options(**kw))
    S = dict(n=lay.n, m=lay.m, num_linear=lay.num_linear, jrow=lay.jrow, jcol=lay.jcol, hrow=lay.hrow, hcol=lay.hcol, xL=lay.xL,
             xU=lay.xU, gL=lay.gL, gU=lay.gU)
    oseats[b] = T._oracle_qp(None, S, O.default_options(kkt_mode=2, **kw))
ctx.sqp_reset()
nlog = {b: 0 for b in ids}
worst = {}
for it in range(iters):
    ctx.sqp_run(1)
    for b in ids:
        log = ctx.sqp_qp_log(b)
        if len(log) == nlog[b]:
            continue
        nlog[b] = len(log)
        rq = ctx.sqp_last_request(b)
        args = (rq["mode"], rq["x_k"], rq["delta"], rq["mu_pen"], rq["c"], rq["b"], rq["jac_coo"], rq["hess_coo"])
        rg = seats[b].qp_solve(*args)
        ro = oseats[b](*args)
        same_log = (rg["status"], rg["ipm_iters"], rg["n_factor"]) == tuple(log[-1][1:])
        solved = ro["status"] == O.MOI_LOCALLY_SOLVED and rg["status"] == ro["status"]
        dp = T.rel(rg["p"], ro["p"]) if solved else float("nan")
        dl = T.rel(rg["lam"], ro["lam"]) if solved else float("nan")
        # objective of the sub-problem at the two solutions: c'p + p'Hp/2 is not available here without H; use the slack sum (FR) and c'p
        vs = abs(rg["slack"].sum() - ro["slack"].sum()) / max(1.0, abs(ro["slack"].sum())) if solved else float("nan")
        cp = abs(rq["c"] @ rg["p"] - rq["c"] @ ro["p"]) / max(1.0, abs(rq["c"] @ ro["p"])) if solved else float("nan")
        print(f"it {it + 1:2d} scen {b:3d} mode {rq['mode']} status {rg['status']}/{ro['status']} log==seat {int(same_log)} ipm {rg['ipm_iters']:3d}/{ro['ipm_iters']:3d} "
              f"fac {rg['n_factor']:3d}/{ro['n_factor']:3d} |dp| {dp:.1e} |dlam| {dl:.1e} |dslack| {vs:.1e} |dc'p| {cp:.1e}", flush=True)
