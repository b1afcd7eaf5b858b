"""Batched SQP-TR run for profiling: python scripts/gpu_sqp_run.py CASE BATCH STEPS KKT_MODE [key=value option ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, contingency, CASES

case, B, steps, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
kw = {}
for a in sys.argv[5:]:
    k, v = a.split("=")
    kw[k] = float(v) if "." in v or "e" in v else int(v)
nb, ng, nl, seed = CASES[case]
base = acopf_synth(nb, ng, nl, seed); lay0 = acopf_layout(base)
okw = dict(kkt_mode=mode, max_iter=3000, tol_infeas=1e-6, tol_residual=1e-4, use_soc=1)
okw.update(kw)
t0 = time.time()
ctx = pkg.Context(lay0.n, lay0.m, lay0.num_linear, lay0.jrow, lay0.jcol, lay0.hrow, lay0.hcol, lay0.xL, lay0.xU, lay0.gL,
                  lay0.gU, pkg.default_options(**okw), batch=B)
ctx.acopf_attach(base, lay0)
for b in range(B):
    net = base if b == 0 else contingency(base, b, seed)
    ctx.acopf_set_instance(b, net, acopf_layout(net))
t1 = time.time()
ctx.sqp_reset(); ctx.sqp_run(1); ctx.reset_counters()
t2 = time.time(); ctx.sqp_run(steps); dt = time.time() - t2
c = ctx.counters()
ret, it, done = ctx.sqp_status()
print(f"[run] {case} B={B} kkt_mode={mode} {kw}: setup {t1 - t0:.2f}s; {dt * 1e3:.1f} ms for {steps} steps, n_qp {c['n_qp']} "
      f"ipm {c['n_ipm_iter']} fac {c['n_factor']} => {c['n_qp'] / dt:.0f} QP/s; sparse={c['sparse']} order {c['kkt_order']} "
      f"sn {c['n_supernodes']} levels {c['n_levels']} maxfront {c['max_front']} nnzL {c['nnz_l']} "
      f"launches f{c['factor_launches']} s{c['solve_launches']}; done {int(np.sum(done))} ret {dict(zip(*np.unique(ret, return_counts=True)))}",
      flush=True)
ctx.close()
