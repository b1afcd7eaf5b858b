"""Per-mode interior-point work of the bench workload in the oracle (ORA_QP_LOG lines of oracle/sqp_tr.c): sub-problems,
iterations and factorisations per solve by mode, split by whether the outer iteration before was accepted.
usage: ipm_mode_stats.py [nscen=32] [iters=25] [lq=1]"""
import os, sys, time, tempfile
from multiprocessing import Pool
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
MI = int(sys.argv[2]) if len(sys.argv) > 2 else 25
LQ = int(sys.argv[3]) if len(sys.argv) > 3 else 1
CASE = sys.argv[4] if len(sys.argv) > 4 else "case118"

def run(s):
    path = f"/tmp/exp/qplog_{os.getpid()}_{s}.txt"
    if os.path.exists(path): os.remove(path)
    os.environ["ORA_QP_LOG"] = path
    from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, contingency, CASES
    from oracle import oracle as O
    nb, ng, nl, seed = CASES[CASE]
    base = acopf_synth(nb, ng, nl, seed)
    net = base if s == 0 else contingency(base, s, seed)
    lay = acopf_layout(net)
    kw = dict(max_iter=MI, tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, literal_quirks=LQ)
    r = O.sqp_solve(O.problem_acopf(net, lay), O.default_options(kkt_mode=2, num_threads=1, ipm_corrector=int(os.environ.get("EXP_CORRECTOR", "1")), **kw))
    rows = np.loadtxt(path, ndmin=2)
    os.remove(path)
    return s, r["status"], r["iter"], rows

if __name__ == "__main__":
    os.makedirs("/tmp/exp", exist_ok=True)
    rng = np.random.default_rng(11)
    scen = sorted(rng.choice(512, size=N, replace=False).tolist())
    t0 = time.time()
    with Pool(8) as p:
        res = p.map(run, scen)
    rows = np.concatenate([r[3] for r in res])
    tot = rows[:, 3].sum()
    print(f"{N} scenarios x {MI} iterations lq={LQ}: {len(rows)} sub-problems, {rows[:,2].mean():.2f} it / {rows[:,3].mean():.2f} fac per solve ({time.time()-t0:.0f} s)")
    for mode, name in ((0, "QP"), (1, "FR"), (2, "SOC"), (3, "LP")):
        r = rows[rows[:, 0] == mode]
        if len(r) == 0: continue
        print(f"  {name:3s}: {len(r):5d} solves  {r[:,2].mean():6.2f} it  {r[:,3].mean():6.2f} fac   {100*r[:,3].sum()/tot:5.1f} % of factorisations   status {sorted(set(r[:,1].astype(int)))}")
        if mode == 0:
            for acc in (1, 0):
                q = r[r[:, 6] == acc]
                if len(q): print(f"       after an {'accepted' if acc else 'rejected'} step: {len(q):5d} solves {q[:,2].mean():6.2f} it {q[:,3].mean():6.2f} fac")
    for mode, name in ((0, "QP"), (2, "SOC")):
        r = rows[rows[:, 0] == mode]
        for st in sorted(set(r[:, 1].astype(int))):
            q = r[r[:, 1] == st]
            print(f"  {name} status {st}: {len(q)} solves {q[:,2].mean():.2f} it {q[:,3].mean():.2f} fac")
