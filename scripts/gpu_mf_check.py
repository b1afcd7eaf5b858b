"""First-light check of the multifrontal path on the GPU box:
 1. kernel level: sqphip_mf_solve_test (device) against sqphip_mf_host_solve (host reference of the same plan)
 2. sub-problem level: every mode on case14, sparse vs dense solver
 3. batched SQP-TR, 64 x case118, a few outer iterations: sparse vs dense, time and counters
Usage: python scripts/gpu_mf_check.py [stage ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, contingency, CASES
from oracle import oracle as O

stages = sys.argv[1:] or ["kernel", "qp", "sqp"]

def mk_ctx(lay, batch=1, **kw):
    return pkg.Context(lay.n, lay.m, lay.num_linear, lay.jrow, lay.jcol, lay.hrow, lay.hcol, lay.xL, lay.xU, lay.gL,
                       lay.gU, pkg.default_options(**kw), batch=batch)

def rel(a, b):
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))

if "kernel" in stages:
    for case, cond in (("case14", 1), ("case14", 0), ("case118", 1), ("case118", 0), ("case1354", 1)):
        nb, ng, nl, seed = CASES[case]
        lay = acopf_layout(acopf_synth(nb, ng, nl, seed))
        n, m = lay.n, lay.m
        rng = np.random.default_rng(1)
        Jv = rng.normal(size=len(lay.jrow)); Hv = 0.1 * rng.normal(size=len(lay.hrow))
        eq = lay.gL == lay.gU
        Dd = rng.uniform(0.1, 10, m); Dd[eq] = rng.uniform(0, 1e-3, eq.sum())
        sigp = rng.uniform(1, 20, n); hd = rng.uniform(0, 1, n)
        rt = np.ones(m, dtype=np.int32); rt[rng.uniform(size=m) < 0.1] = 0
        mk = int(eq.sum()) if cond else m
        rhs = rng.normal(size=n + mk)
        t0 = time.time()
        ctx = mk_ctx(lay, batch=3, kkt_mode=2, kkt_condense=cond)
        t1 = time.time()
        c = ctx.counters()
        ref, dref, npos = pkg.mf_host_solve(n, m, lay.jrow, lay.jcol, lay.hrow, lay.hcol, lay.gL, lay.gU, cond, Jv, Hv, Dd,
                                            sigp, hd, rt, 0.7, 1e-3, rhs)
        for inst in (0, 2):
            a, b, dv = ctx.mf_solve_test(inst, Jv, Hv, Dd, sigp, hd, rt, 0.7, 1e-3, rhs)
            print(f"[kernel] {case} cond={cond} inst={inst}: order {c['kkt_order']} sn {c['n_supernodes']} lev {c['n_levels']} "
                  f"maxfront {c['max_front']} nnzL {c['nnz_l']}: fused {rel(a, ref):.1e} standalone {rel(b, ref):.1e} "
                  f"dinv {rel(dv, dref):.1e} npos {int((dv > 0).sum())}/{npos} (create {t1 - t0:.2f}s)", flush=True)
        ctx.close()

if "qp" in stages:
    nb, ng, nl, seed = CASES["case14"]
    net = acopf_synth(nb, ng, nl, seed); lay = acopf_layout(net)
    P = O.problem_acopf(net, lay)
    cs = mk_ctx(lay, kkt_mode=2); cd = mk_ctx(lay, kkt_mode=1)
    rng = np.random.default_rng(2)
    xr = np.clip(lay.x0 + 0.02 * rng.standard_normal(lay.n), lay.xL, lay.xU)
    for x, lam in ((lay.x0, np.zeros(lay.m)), (xr, 50 * rng.standard_normal(lay.m))):
        df, E, jv, hv = P.eval_grad_f(x), P.eval_g(x), P.eval_jac_g(x), P.eval_h(x, 1.0, lam)
        for mode, delta in ((O.MODE_LP, 10.0), (O.MODE_QP, 10.0), (O.MODE_QP, 0.2), (O.MODE_FR, 0.2), (O.MODE_SOC, 1.0),
                            (O.MODE_L1QP, 1.0), (O.MODE_INFEAS, 1.0)):
            a = cs.qp_solve(mode, x, delta, 3.0, df, E, jv, hv); b = cd.qp_solve(mode, x, delta, 3.0, df, E, jv, hv)
            print(f"[qp] mode {mode} delta {delta}: status {a['status']}/{b['status']} iters {a['ipm_iters']}/{b['ipm_iters']} "
                  f"fac {a['n_factor']}/{b['n_factor']} |dp| {np.abs(a['p'] - b['p']).max():.1e} "
                  f"|dlam| {np.abs(a['lam'] - b['lam']).max():.1e}", flush=True)
    cs.close(); cd.close()

if "sqp" in stages:
    for case, B, steps in (("case14", 8, 8), ("case118", 64, 6)):
        nb, ng, nl, seed = CASES[case]
        base = acopf_synth(nb, ng, nl, seed); lay0 = acopf_layout(base)
        res = {}
        for mode in (2, 1):
            ctx = mk_ctx(lay0, batch=B, kkt_mode=mode, max_iter=3000, tol_infeas=1e-6, tol_residual=1e-4, use_soc=1)
            ctx.acopf_attach(base, lay0)
            for b in range(B):
                net = base if b == 0 else contingency(base, b, seed)
                ctx.acopf_set_instance(b, net, acopf_layout(net))
            ctx.sqp_reset(); ctx.sqp_run(1); ctx.reset_counters()
            t0 = time.time(); ctx.sqp_run(steps); dt = time.time() - t0
            c = ctx.counters()
            xs = np.stack([ctx.sqp_get(b)["x"] for b in range(B)])
            st = [(ctx.sqp_get(b)["status"], ctx.sqp_get(b)["iter"]) for b in range(B)]
            res[mode] = (xs, st)
            print(f"[sqp] {case} B={B} kkt_mode={mode}: {dt * 1e3:.1f} ms for {steps} steps, n_qp {c['n_qp']} ipm {c['n_ipm_iter']} "
                  f"fac {c['n_factor']}  => {c['n_qp'] / dt:.0f} QP/s; sparse={c['sparse']} order {c['kkt_order']}", flush=True)
            ctx.close()
        print(f"[sqp] {case}: |x_sparse - x_dense| max {np.abs(res[2][0] - res[1][0]).max():.2e}; statuses equal "
              f"{res[2][1] == res[1][1]}", flush=True)
