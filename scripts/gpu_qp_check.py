"""GPU sanity: QP sub-problem and batched SQP parity against the CPU oracle."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, contingency, CASES
from oracle import oracle as O

def rel(a, b):
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))

def qp_inputs(P, S, x, lam):
    jv = P.eval_jac_g(x); hv = P.eval_h(x, 1.0, lam) if S["hrow"].size else None
    return P.eval_grad_f(x), P.eval_g(x), jv, hv

def oracle_qp(P, S, opts=None):
    n, m = S["n"], S["m"]
    jcp, jrv, jslot, _ = O.coo_to_csc(n, S["jrow"], S["jcol"])
    hcp, hrv, hslot, hslot_t = O.coo_to_csc(n, S["hrow"], S["hcol"], sym=True)
    q = O.QpSolver(n, m, S["num_linear"], jcp, jrv, hcp, hrv, S["xL"], S["xU"], S["gL"], S["gU"], opts)
    def solve(mode, x, delta, mu, df, E, jcoo, hcoo):
        jv = np.zeros(len(jrv)); np.add.at(jv, jslot, jcoo)
        hv = np.zeros(len(hrv))
        if hcoo is not None:
            np.add.at(hv, hslot, hcoo); ok = hslot_t >= 0; np.add.at(hv, hslot_t[ok], hcoo[ok])
        return q.solve(mode, x, delta, mu, df, E, jv, hv, want_slack=True)
    return solve

which = sys.argv[1:] or ["toy", "hs071", "case14", "sqp14"]
for name in which:
    if name in ("toy", "hs071", "readme1"):
        P = getattr(O, "problem_" + name)(); S = P.structure()
        ctx = pkg.Context(S["n"], S["m"], S["num_linear"], S["jrow"], S["jcol"], S["hrow"], S["hcol"],
                          S["xL"], S["xU"], S["gL"], S["gU"])
        osolve = oracle_qp(P, S)
        rng = np.random.default_rng(1)
        for trial in range(4):
            x = P.x0 + (0.3 * rng.standard_normal(S["n"]) if trial else 0)
            x = np.clip(x, np.maximum(S["xL"], -1e3), np.minimum(S["xU"], 1e3))
            lam = rng.standard_normal(S["m"]) * (trial > 0)
            df, E, jv, hv = qp_inputs(P, S, x, lam)
            for mode in (O.MODE_QP, O.MODE_FR, O.MODE_LP, O.MODE_L1QP, O.MODE_INFEAS):
                for delta in (10.0, 0.5):
                    ro = osolve(mode, x, delta, 7.0, df, E, jv, hv)
                    rg = ctx.qp_solve(mode, x, delta, 7.0, df, E, jv, hv)
                    print(name, "trial", trial, "mode", mode, "delta", delta, "status", ro["status"], rg["status"],
                          "ipm", ro["ipm_iters"], rg["ipm_iters"], "dp %.1e dlam %.1e dU %.1e dL %.1e" % (
                          rel(rg["p"], ro["p"]), rel(rg["lam"], ro["lam"]), rel(rg["mult_x_U"], ro["mult_x_U"]),
                          rel(rg["mult_x_L"], ro["mult_x_L"])), flush=True)
    elif name.startswith("case"):
        nb, ng, nl, seed = CASES[name]
        net = acopf_synth(nb, ng, nl, seed); lay = acopf_layout(net)
        P = O.problem_acopf(net, lay); S = P.structure()
        ctx = pkg.Context(lay.n, lay.m, lay.num_linear, lay.jrow, lay.jcol, lay.hrow, lay.hcol,
                          lay.xL, lay.xU, lay.gL, lay.gU, batch=1)
        ctx.acopf_attach(net, lay); ctx.acopf_set_instance(0, net, lay)
        rng = np.random.default_rng(2)
        x = lay.x0 + 0.02 * rng.standard_normal(lay.n); lam = rng.standard_normal(lay.m)
        ev = ctx.acopf_eval(0, x, 1.0, lam)
        print(name, "eval: f %.1e grad %.1e g %.1e jac %.1e hess %.1e" % (
            abs(ev["f"] - P.eval_f(x)) / abs(P.eval_f(x)), rel(ev["grad"], P.eval_grad_f(x)), rel(ev["g"], P.eval_g(x)),
            rel(ev["jval"], P.eval_jac_g(x)), rel(ev["hval"], P.eval_h(x, 1.0, lam))), flush=True)
        osolve = oracle_qp(P, S, O.default_options(num_threads=8))
        for trial in range(2):
            xx = lay.x0 if trial == 0 else np.clip(x, lay.xL, lay.xU)
            ll = np.zeros(lay.m) if trial == 0 else 50 * lam
            df, E, jv, hv = qp_inputs(P, S, xx, ll)
            for mode, delta in ((O.MODE_LP, 10.0), (O.MODE_QP, 10.0), (O.MODE_QP, 0.2), (O.MODE_FR, 0.2), (O.MODE_SOC, 1.0)):
                t0 = time.time(); ro = osolve(mode, xx, delta, 3.0, df, E, jv, hv); t1 = time.time()
                rg = ctx.qp_solve(mode, xx, delta, 3.0, df, E, jv, hv); t2 = time.time()
                print(name, "trial", trial, "mode", mode, "delta", delta, "status", ro["status"], rg["status"],
                      "ipm", ro["ipm_iters"], rg["ipm_iters"], "fac", ro["n_factor"], rg["n_factor"],
                      "dp %.1e dlam %.1e dU %.1e dL %.1e" % (
                      rel(rg["p"], ro["p"]), rel(rg["lam"], ro["lam"]), rel(rg["mult_x_U"], ro["mult_x_U"]),
                      rel(rg["mult_x_L"], ro["mult_x_L"])), "t_cpu %.2fs t_gpu %.2fs" % (t1 - t0, t2 - t1), flush=True)
    elif name.startswith("sqp"):
        case = "case" + name[3:]
        nb, ng, nl, seed = CASES[case]
        base = acopf_synth(nb, ng, nl, seed)
        B = 4
        nets = [base] + [contingency(base, s, seed) for s in range(1, B)]
        lays = [acopf_layout(nt) for nt in nets]
        lay = lays[0]
        for quirks, iters in ((1, 12), (0, 40)):
            opts = pkg.default_options(max_iter=iters, tol_infeas=1e-6, tol_residual=1e-4, literal_quirks=quirks)
            ctx = pkg.Context(lay.n, lay.m, lay.num_linear, lay.jrow, lay.jcol, lay.hrow, lay.hcol,
                              lay.xL, lay.xU, lay.gL, lay.gU, opts, batch=B)
            ctx.acopf_attach(base, lay)
            for b in range(B): ctx.acopf_set_instance(b, nets[b], lays[b])
            ctx.sqp_reset(); t0 = time.time(); ctx.sqp_run(0); tg = time.time() - t0
            for b in range(B):
                P = O.problem_acopf(nets[b], lays[b])
                oo = O.default_options(max_iter=iters, tol_infeas=1e-6, tol_residual=1e-4, literal_quirks=quirks, num_threads=8)
                t0 = time.time(); ro = O.sqp_solve(P, oo); tc = time.time() - t0
                rg = ctx.sqp_get(b); tr = ctx.sqp_trace(b)
                nmatch = 0
                for a, c in zip(ro["trace"], tr):
                    if (a["iter"], a["accepted"], a["fr"], a["sub_status"]) == (c["iter"], c["accepted"], c["fr"], c["sub_status"]) and abs(a["delta"] - c["delta"]) <= 1e-9 * max(1, abs(a["delta"])): nmatch += 1
                    else: break
                print(f"{name} quirks={quirks} inst {b}: status {ro['status']} {rg['status']} iter {ro['iter']} {rg['iter']} "
                      f"trace rows {len(ro['trace'])} {len(tr)} matching-prefix {nmatch} dx {rel(rg['x'], ro['x']):.1e} "
                      f"dmult_g {rel(rg['mult_g'], ro['mult_g']):.1e} dobj {abs(rg['obj_val']-ro['obj_val'])/abs(ro['obj_val']):.1e} "
                      f"t_cpu {tc:.2f}s", flush=True)
            print(f"  gpu batch time {tg:.2f}s counters", ctx.counters(), flush=True)
