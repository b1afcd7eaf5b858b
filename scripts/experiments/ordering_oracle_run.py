import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, "/root/repo")
from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, contingency, CASES
from oracle import oracle as O
case = sys.argv[1] if len(sys.argv) > 1 else "case118"
mi = int(sys.argv[2]) if len(sys.argv) > 2 else 25
lq = int(sys.argv[3]) if len(sys.argv) > 3 else 1
scen = [int(s) for s in (sys.argv[4].split(",") if len(sys.argv) > 4 else ["0", "7", "3"])]
nb, ng, nl, seed = CASES[case]
base = acopf_synth(nb, ng, nl, seed)
L = O.lib()
out = (C.c_double * 6)()
for s in scen:
    net = base if s == 0 else contingency(base, s, seed)
    lay = acopf_layout(net)
    kw = dict(max_iter=mi, tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, literal_quirks=lq)
    t0 = time.time()
    r = O.sqp_solve(O.problem_acopf(net, lay), O.default_options(kkt_mode=2, num_threads=1, **kw))
    dt = time.time() - t0
    L.ora_dbg_counts(out)
    tr = r["trace"]
    print(f"ROWS_AFTER={os.environ.get('ORA_ROWS_AFTER','1')} lq={lq} scen {s}: status {r['status']} iter {r['iter']} f {r['obj_val'] if 'obj_val' in r else r.get('f')} ipm {sum(t['ipm_iters'] for t in tr)} "
          f"solves {out[0]:.0f} refines {out[1]:.0f} maxres0 {out[2]:.1e} maxres1 {out[3]:.1e} bad0 {out[4]:.0f} bad1 {out[5]:.0f}  {dt:.1f}s", flush=True)
    np.save(f"/tmp/exp/x_{case}_{s}_{lq}_{os.environ.get('ORA_ROWS_AFTER','1')}.npy", r["x"])
