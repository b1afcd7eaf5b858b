"""Interior-point work on the bench workload, in the oracle: N scenarios x 25 outer iterations of the 512 x IEEE-118 job
(example SQP options, reference Hessian sign), one scenario per process.  Prints sub-problems, interior-point iterations
and factorisations per sub-problem, overall and by convexity (a sub-problem whose solve needed an inertia correction
counts as non-convex: n_factor > ipm_iters).  Policies are switched by ORA_* environment variables (oracle/qp_ipm.c).
usage: ipm_policy_run.py [nscen=32] [iters=25] [lq=1] [case=case118]"""
import os, sys, time
from multiprocessing import Pool
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
MI = int(sys.argv[2]) if len(sys.argv) > 2 else 25
LQ = int(sys.argv[3]) if len(sys.argv) > 3 else 1
CASE = sys.argv[4] if len(sys.argv) > 4 else "case118"

def run(s):
    from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, contingency, CASES
    from oracle import oracle as O
    nb, ng, nl, seed = CASES[CASE]
    base = acopf_synth(nb, ng, nl, seed)
    net = base if s == 0 else contingency(base, s, seed)
    lay = acopf_layout(net)
    kw = dict(max_iter=MI, tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, literal_quirks=LQ)
    r = O.sqp_solve(O.problem_acopf(net, lay), O.default_options(kkt_mode=2, num_threads=1, **kw))
    rows = [(t["ipm_iters"], t["n_factor"], t["fr"], t["sub_status"]) for t in r["trace"]]
    return s, r["status"], r["iter"], r["n_qp"], r["n_ipm_iter"], r["n_factor"], float(r["obj_val"]), rows

if __name__ == "__main__":
    rng = np.random.default_rng(11)
    scen = sorted(rng.choice(512, size=N, replace=False).tolist())
    t0 = time.time()
    with Pool(8) as p:
        res = p.map(run, scen)
    nqp = sum(r[3] for r in res); it = sum(r[4] for r in res); fac = sum(r[5] for r in res)
    print(f"{N} scenarios x {MI} iterations lq={LQ}: {nqp} sub-problems, {it / nqp:.2f} iterations / {fac / nqp:.2f} factorisations per sub-problem"
          f"   ({time.time() - t0:.0f} s)   status {sorted(set(r[1] for r in res))}  iters {sum(r[2] for r in res)}")
    tag = os.environ.get("EXP_TAG")
    if tag:
        np.save(f"/tmp/exp/{tag}.npy", np.array([(r[0], r[1], r[2], r[6]) for r in res]))
