"""Parity in depth (run with -m gpu on an MI355X; VERDICT r2 "what's weak" 2-4): what the oracle and the device can be
held to on the configuration the bench times (reference Hessian sign, 512 resident scenarios), on the large shapes and
on networks no other test pins.  Everything goes through the C ABI of libsqphip.so; the oracle (oracle/) is the checker.

The invariant on the headline workload.  With the reference's JuMP-sign Hessian the sub-problems
are non-convex and every run passes through degenerate feasibility-restoration LPs whose optimal face is not a point: two
correct solvers leave such an LP at different points and the trajectories separate (measured with
scripts/gpu_decisions_depth.py: 4 of 12 random bench scenarios take a different accept / reject decision somewhere
between iterations 13 and 23, |dx| up to 0.18 at iteration 25).  What does hold, and is asserted here, is parity of
every sub-problem along the device's own trajectory: the request the batched run worked on is fetched from the device
and solved again by the drop-in seat of a fresh context (must reproduce the batched run's status and work counters
exactly) and by the oracle's seat (same status; step and multipliers at 1e-8 / 1e-6 for the trust-region and correction
QPs; the optimal VALUE for restoration LPs).  Data behind the tolerances: scripts/gpu_replay_depth.py, 200 sub-problems:
status 200 / 200, counters 196 / 200 equal, max |dp| 4.3e-9, max |dlambda| 1.1e-8, restoration optimal value 1.9e-7."""
import os

import numpy as np
import pytest

import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, acr_layout, contingency, synth_case, CASES
from oracle import oracle as O
from test_gpu_parity import _ipm_counts_close, _oracle_qp, _run_batch, _same_decisions, host_threads, rel, TOL, GOLD

pytestmark = pytest.mark.gpu


def _structure(lay):
    return dict(n=lay.n, m=lay.m, num_linear=lay.num_linear, jrow=lay.jrow, jcol=lay.jcol, hrow=lay.hrow, hcol=lay.hcol,
                xL=lay.xL, xU=lay.xU, gL=lay.gL, gU=lay.gU)


def _ctx(lay, opts, batch=1):
    return pkg.Context(lay.n, lay.m, lay.num_linear, lay.jrow, lay.jcol, lay.hrow, lay.hcol, lay.xL, lay.xU, lay.gL, lay.gU,
                       opts, batch=batch)


def test_subproblems_of_the_bench_run_replay_through_both_seats():
    """The bench configuration itself: 512 resident IEEE-118-shaped scenarios, the example's SQP options, the reference's
    Hessian sign.  Eight scenarios drawn at random, 25 outer iterations: after every iteration the sub-problem each of
    them worked on last (trust-region QP, second-order correction or restoration LP) is replayed -- see the module text."""
    nb, ng, nl, seed = CASES["case118"]
    base = acopf_synth(nb, ng, nl, seed)
    lay0 = acopf_layout(base)
    B, iters = 512, 25
    kw = dict(max_iter=3000, tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, literal_quirks=1)
    ctx = _ctx(lay0, pkg.default_options(**kw), batch=B)
    ctx.acopf_attach(base, lay0)
    lays = {}
    for b in range(B):
        net = base if b == 0 else contingency(base, b, seed)
        lays[b] = acopf_layout(net)
        ctx.acopf_set_instance(b, net, lays[b])
    ids = sorted(np.random.default_rng(5).choice(B, size=8, replace=False).tolist())
    seats = {b: _ctx(lays[b], pkg.default_options(**kw)) for b in ids}
    oseats = {b: _oracle_qp(None, _structure(lays[b]), O.default_options(kkt_mode=2, **kw)) for b in ids}
    ctx.sqp_reset()
    nlog = {b: 0 for b in ids}
    n_sub = n_equal_counts = 0
    worst_fr = 0.0
    worst_mult = 0.0
    modes = set()
    for _ in range(iters):
        ctx.sqp_run(1)
        for b in ids:
            log = ctx.sqp_qp_log(b)
            if len(log) == nlog[b]:
                continue
            nlog[b] = len(log)
            rq = ctx.sqp_last_request(b)
            args = (rq["mode"], rq["x_k"], rq["delta"], rq["mu_pen"], rq["c"], rq["b"], rq["jac_coo"], rq["hess_coo"])
            rg, ro = seats[b].qp_solve(*args), oseats[b](*args)
            # the batched run and the seat are one code path: same status, same work, bit for bit
            assert (rg["status"], rg["ipm_iters"], rg["n_factor"]) == tuple(log[-1][1:]), (b, log[-1])
            assert rg["status"] == ro["status"], (b, rq["mode"])
            n_sub += 1
            modes.add(rq["mode"])
            n_equal_counts += (rg["ipm_iters"], rg["n_factor"]) == (ro["ipm_iters"], ro["n_factor"])
            if rq["mode"] == O.MODE_FR:
                # a degenerate LP: the iterate wanders along the optimal face until the error measure crosses the
                # threshold, and the last digits of the Newton directions decide when (seen: 29 against 21 on one of
                # 200); held to the optimal value below and to the 95 % rule at the end, loosely here
                worst_fr = max(worst_fr, abs(rg["ipm_iters"] - ro["ipm_iters"]) / ro["ipm_iters"])
                assert abs(rg["ipm_iters"] - ro["ipm_iters"]) <= max(2, 0.5 * ro["ipm_iters"]), (b, rq["mode"], rg["ipm_iters"], ro["ipm_iters"])
            else:
                assert abs(rg["ipm_iters"] - ro["ipm_iters"]) <= max(2, 0.2 * ro["ipm_iters"]), (b, rq["mode"], rg["ipm_iters"], ro["ipm_iters"])
            if ro["status"] != O.MOI_LOCALLY_SOLVED:
                continue
            if rq["mode"] == O.MODE_FR:          # a linear programme: the optimal value is what is unique
                vo, vg = float(ro["slack"].sum()), float(rg["slack"].sum())
                assert abs(vg - vo) <= 1e-6 * max(1.0, abs(vo)), (b, vg, vo)
            else:
                assert rel(rg["p"], ro["p"]) < TOL, (b, rq["mode"])
                # multipliers: the least determined output of an interior-point solve that stops on a scaled error of 1e-9
                # (a bound multiplier is mu / gap of the last iterate).  Data: two collections of 200 sub-problems along
                # two different trajectories of this run (the rounding of the solve kernels changed between them): row
                # multipliers at most 1.1e-8 and 9.0e-8, reduced costs at most 1.0e-8 and 2.1e-7 of the largest one
                # (round 4: the monotone barrier rule is the default -- it ends on well-centred iterates -- and the bound is the
                #  5e-7 the verdict asked for; the worst case of this run is printed below)
                e_lam, e_rc = rel(rg["lam"], ro["lam"]), rel(rg["mult_x_L"] - rg["mult_x_U"], ro["mult_x_L"] - ro["mult_x_U"])
                worst_mult = max(worst_mult, e_lam, e_rc)
                assert e_lam < 5e-7 and e_rc < 5e-7, (b, rq["mode"], e_lam, e_rc)
    assert n_sub >= 8 * iters - 8 and {O.MODE_QP, O.MODE_SOC, O.MODE_FR} <= modes
    # equal interior-point iteration AND factorisation counts on all but a few sub-problems (degenerate LPs, one
    # infeasible QP in the collection run): two implementations with different elimination orders
    assert n_equal_counts >= 0.95 * n_sub, (n_equal_counts, n_sub)
    print(f"replay: {n_sub} sub-problems, {n_equal_counts} with equal counts, worst restoration-LP count gap {worst_fr:.2f}, worst multiplier error {worst_mult:.1e}")
    for c in seats.values():
        c.close()
    ctx.close()


def test_case118_prefix_before_the_first_restoration_is_exact():
    """The same workload up to the iteration in front of its first restoration LP (three outer iterations: linear phase,
    an accepted trust-region step, a rejected one with its second-order correction, the infeasible QP that enters
    restoration): nothing degenerate has been solved yet, so the whole state must agree -- every decision, the iterate
    and the objective at 1e-8.  Interior-point iteration counts are held to the allowance of test_gpu_parity.py (two
    iterations or 20 %): a solve ends when its error measure crosses a threshold; the base case's first QP crosses it at
    iteration 19, 20 or 21 depending on the elimination order (oracle dense / device / oracle sparse with every
    direction refined), contingency 7's correction at 14 or 17, and forcing a refinement step on every direction on both
    sides (SQPHIP_REFINE_TOL = ORA_REFINE_TOL = 0) does not remove that (measured: 20 against 21) -- equal counts are
    asserted statistically instead, over the 200 sub-problems of the replay test above."""
    nb, ng, nl, seed = CASES["case118"]
    base = acopf_synth(nb, ng, nl, seed)
    nets = [base, contingency(base, 7, seed), contingency(base, 3, seed), contingency(base, 262, seed)]
    lays = [acopf_layout(nt) for nt in nets]
    kw = dict(max_iter=3, tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, literal_quirks=1)
    ctx = _run_batch(nets, lays, kw)
    for b in range(len(nets)):
        ro = O.sqp_solve(O.problem_acopf(nets[b], lays[b]), O.default_options(kkt_mode=2, **kw))
        rg, tr = ctx.sqp_get(b), ctx.sqp_trace(b)
        assert (rg["status"], rg["iter"]) == (ro["status"], ro["iter"])
        assert _same_decisions(ro, tr)
        assert _ipm_counts_close(ro, tr), b
        assert rel(rg["x"], ro["x"]) < TOL and abs(rg["obj_val"] - ro["obj_val"]) <= TOL * abs(ro["obj_val"]), b
    ctx.close()


def test_whole_subproblems_on_case9241_match_oracle():
    """BASELINE.json configs[4] (9241pegase shape, Newton matrix of order 232 443 condensed to 168 247, 6.95 M entries in
    the fronts): two whole sub-problems at the start point through `sqphip_qp_solve` against the oracle's independent
    sparse LDL^T -- the linear-phase projection QP (strictly convex: status, iteration count, the step at 1e-8) and the
    restoration LP (status; its optimal value when solved)."""
    nb, ng, nl, seed = CASES["case9241"]
    net = acopf_synth(nb, ng, nl, seed); lay = acopf_layout(net)
    P = O.problem_acopf(net, lay)
    ctx = _ctx(lay, pkg.default_options())
    assert ctx.counters()["sparse"] == 1 and ctx.counters()["kkt_order"] == 168247
    osolve = _oracle_qp(P, P.structure(), O.default_options(kkt_mode=2, num_threads=16))
    x = lay.x0
    df, E, jv, hv = P.eval_grad_f(x), P.eval_g(x), P.eval_jac_g(x), P.eval_h(x, 1.0, np.zeros(lay.m))
    for mode in (O.MODE_LP, O.MODE_FR):
        ro, rg = osolve(mode, x, 10.0, 1.0, df, E, jv, hv), ctx.qp_solve(mode, x, 10.0, 1.0, df, E, jv, hv)
        assert rg["status"] == ro["status"], mode
        assert abs(rg["ipm_iters"] - ro["ipm_iters"]) <= max(2, 0.2 * ro["ipm_iters"]), mode
        if ro["status"] == O.MOI_LOCALLY_SOLVED:
            if mode == O.MODE_LP:
                assert rel(rg["p"], ro["p"]) < TOL and rel(rg["lam"], ro["lam"]) < 1e-6
            else:
                assert abs(rg["slack"].sum() - ro["slack"].sum()) <= 1e-5 * max(1.0, ro["slack"].sum())
    ctx.close()


def test_case9241_line_outage_scenarios_get_past_their_flat_qps():
    """BASELINE.json configs[4] as bench.py runs it (geographic 9241-bus network, line-outage scenarios, textbook sign).
    The trust-region QP behind the linear phase has nearly flat directions; the regularised Newton iteration stalls at an
    error of 1e-6 ... 4e-6 there (oracle log of scenario 1: iterations 60 - 200).  Before the third acceptable-termination
    rule (25 iterates within 10^4 x ipm_tol; DESIGN.md section 3, same in oracle/qp_ipm.c and ipm.hip) those QPs ran into
    the 200-iteration limit and run! stopped 98 of 128 scenarios after two outer iterations.  Oracle, scenario 1 with the
    rule: QPs of 84 and 89 iterations, restoration, then 25 - 29 iterations per QP.  The oracle takes minutes per
    sub-problem at this size, so the device is held to the behaviour: four scenarios, five outer iterations, every one
    still running (no sub-problem status that ends the run), no QP at the iteration limit."""
    base = synth_case("case9241")
    seed = CASES["case9241"][3]
    nets = [base] + [contingency(base, s, seed) for s in (1, 2, 3)]
    lays = [acopf_layout(nt) for nt in nets]
    kw = dict(max_iter=5, tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, literal_quirks=0)
    ctx = _run_batch(nets, lays, kw)
    for b in range(len(nets)):
        rg = ctx.sqp_get(b)
        assert rg["iter"] >= 5 and rg["status"] == -1, (b, rg["status"], rg["iter"])       # -1: stopped by max_iter, nothing else
        log = ctx.sqp_qp_log(b)
        assert all(row[2] < 200 for row in log), (b, [row[1:] for row in log])
    ctx.close()


@pytest.mark.parametrize("form", ["polar", "acr"])
def test_unpinned_networks_terminate_like_the_oracle(form):
    """Robustness on networks no other test pins (scripts/gpu_seed_fuzz.py promoted; the script runs seven seeds to 60
    iterations, the test three to 40 to stay near a minute per formulation): other generator seeds of the IEEE-118 shape, base case and two
    contingencies each, polar and rectangular voltages, run to termination (textbook Hessian sign).  Every run must end
    with the oracle's status; converged runs at the same point (1e-6) after the same number of outer iterations --
    except that ONE run of the 9 per formulation may differ by one outer iteration: a termination test decided at its
    threshold (round 2's collection of 42 runs had one: the device converged one iteration later to the same point, 5e-9)."""
    from concurrent.futures import ThreadPoolExecutor
    nb, ng, nl, seed0 = CASES["case118"]
    layout = acopf_layout if form == "polar" else acr_layout
    kw = dict(max_iter=40, tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, literal_quirks=0)
    jobs, dev = [], {}
    for seed in range(seed0 + 1, seed0 + 4):
        base = acopf_synth(nb, ng, nl, seed)
        nets = [base, contingency(base, 1, seed), contingency(base, 2, seed)]
        lays = [layout(nt) for nt in nets]
        ctx = _ctx(lays[0], pkg.default_options(**kw), batch=3)
        ctx.acopf_attach(base, lays[0])
        for b in range(3):
            ctx.acopf_set_instance(b, nets[b], lays[b])
            jobs.append((seed, b, nets[b], lays[b]))
        ctx.sqp_reset(); ctx.sqp_run(0)
        for b in range(3):
            dev[(seed, b)] = ctx.sqp_get(b)
        ctx.close()
    # the oracle runs side by side (ctypes releases the GIL inside the oracle's solver)
    with ThreadPoolExecutor(max_workers=host_threads()) as ex:
        ora = list(ex.map(lambda j: O.sqp_solve(O.problem_acopf(j[2], j[3]), O.default_options(num_threads=1, **kw)), jobs))
    off_by_one = 0
    for (seed, b, _, _), ro in zip(jobs, ora):
        rg = dev[(seed, b)]
        assert rg["status"] == ro["status"], (seed, b)
        if rg["iter"] != ro["iter"]:
            assert abs(rg["iter"] - ro["iter"]) == 1 and ro["status"] == 0, (seed, b, rg["iter"], ro["iter"])
            off_by_one += 1
        if ro["status"] == 0:
            assert rel(rg["x"], ro["x"]) < 1e-6, (seed, b)
            assert abs(rg["obj_val"] - ro["obj_val"]) <= 1e-7 * abs(ro["obj_val"]), (seed, b)
    assert off_by_one <= 1


def test_case1354_geo_converges_like_the_oracle():
    """BASELINE.json configs[2] as a CONVERGENT problem (round 3: `acopf_synth_geo`, the lattice-strip network with local
    generation; the SURVEY.md section 8d recipe stays in restoration at this size): base case and one contingency run to
    convergence on the device (multifrontal path, order 21 865) and in the oracle (its own sparse LDL^T and ordering) --
    status, outer iteration count and every accept / reject / restoration decision equal, objective and point at 1e-8.
    Termination as /root/reference/src/algorithms/sqp_trust_region.jl:187-204; the reference's own example runs its
    ACOPF cases to convergence (examples/acopf/opf.jl:76-84)."""
    base = synth_case("case1354")
    nets = [base, contingency(base, 17, CASES["case1354"][3])]
    lays = [acopf_layout(nt) for nt in nets]
    kw = dict(max_iter=40, tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, literal_quirks=0)
    ctx = _run_batch(nets, lays, kw)
    c = ctx.counters()
    assert c["sparse"] == 1 and c["kkt_order"] == 21865 and c["max_front"] <= 208
    for b in range(2):
        ro = O.sqp_solve(O.problem_acopf(nets[b], lays[b]), O.default_options(kkt_mode=2, num_threads=16, **kw))
        rg, tr = ctx.sqp_get(b), ctx.sqp_trace(b)
        assert ro["status"] == 0 and (rg["status"], rg["iter"]) == (ro["status"], ro["iter"]), b
        assert _same_decisions(ro, tr) and _ipm_counts_close(ro, tr), b
        assert abs(rg["obj_val"] - ro["obj_val"]) <= TOL * abs(ro["obj_val"]) and rel(rg["x"], ro["x"]) < TOL, b
    ctx.close()


def test_queue_shared_between_two_ranks_gives_the_single_rank_results(tmp_path):
    """SURVEY.md section 8f-4 remainder: a scenario queue shared between ranks.  Two fresh rank processes on the one GPU
    of the box (--backend gloo --one-device, as the status-gather rehearsal) run 14 IEEE-14-shaped scenarios through 3
    slots each; rank 0 starts with three quarters of the ids, so rank 1 runs dry and is handed unstarted ids by the
    host-side exchange of sqpsolver.jl_amd/shard.py (run_shared_queue).  Per scenario the two-rank job must return the
    bits of the single-rank run: status, iteration count, objective."""
    import json, subprocess, sys as _sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--workload", "case14", "--batch", "3", "--shared-queue", "14", "--literal-quirks", "0", "--quick"]
    one = tmp_path / "one.json"; two = tmp_path / "two.json"
    r1 = subprocess.run([_sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--dump-status", str(one)] + common,
                        check=True, cwd=root, capture_output=True, timeout=600)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT="29537")
        procs.append(subprocess.Popen([_sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo",
                                       "--one-device", "--queue-split", "0.75", "--dump-status", str(two)] + common, cwd=root,
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-800:] for o in outs]
    line = json.loads(outs[0][0].decode().strip().splitlines()[-1])
    assert line["mode"] == "shared_queue" and line["n_gpus"] == 2 and sum(line["solved_by_rank"]) == 14
    assert line["solved_by_rank"][1] > 14 - round(14 * 0.75)          # rank 1 solved more than its initial share
    a, b = json.load(open(one)), json.load(open(two))
    assert a == b and all(i > 0 for i in a["iter"])
    assert json.loads(r1.stdout.decode().strip().splitlines()[-1])["converged"] >= 12


def test_case9241_scenario_matches_the_oracle_fixture(capsys):
    """BASELINE.json configs[4] as bench.py runs it -- the geographic 9241-bus network, line-outage scenario 1 -- against the
    oracle's fixture tests/golden/geo9241_s1.npz (tests/golden/make_golden_9241.py; the oracle needs ~1.5 minutes per
    sub-problem at this size, so its outputs are committed): (a) the bench's configuration, reference Hessian sign, the
    first seven outer iterations: every decision of the trace and every sub-problem's status exact, radii at 1e-7,
    interior-point counts within two (or 20 %), the objective of the truncated trajectory (its sub-problems are accurate to
    ~1e-6 along nearly flat directions: the distance of the points is reported); (b) the textbook sign
    to convergence: status, outer iterations and decisions exact, objective and point at 1e-8.  The sub-problems this shape
    ends by the third acceptable-termination rule (25 iterates within 1e4 x ipm_tol) are reported with the scaled error
    they were accepted at -- on both sides."""
    path = os.path.join(GOLD, "geo9241_s1.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/geo9241_s1.npz has not been generated (tests/golden/make_golden_9241.py)")
    G = np.load(path)
    base = synth_case("case9241")
    seed = CASES["case9241"][3]
    net = contingency(base, 1, seed); lay = acopf_layout(net)
    report = []
    for tag, kw in (("quirks1_7it", dict(literal_quirks=1, max_iter=7)), ("quirks0_conv", dict(literal_quirks=0, max_iter=60))):
        ctx = _run_batch([net], [lay], dict(tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, **kw))
        rg = ctx.sqp_get(0)
        tr = ctx.sqp_trace(0)
        log, term = ctx.sqp_qp_log(0), ctx.sqp_qp_log_term(0)
        ctx.close()
        gt, gq = G[f"{tag}_trace"], G[f"{tag}_qps"]
        assert (rg["status"], rg["iter"]) == (int(G[f"{tag}_status"]), int(G[f"{tag}_iter"])), tag
        assert [(t["iter"], t["accepted"], t["fr"], t["sub_status"]) for t in tr] == [tuple(int(v) for v in row[:4]) for row in gt], tag
        assert np.allclose([t["delta"] for t in tr], gt[:, 4], rtol=1e-7), tag
        # per sub-problem: mode and status exact, interior-point counts close, the termination rule and its error reported
        assert len(log) == len(gq), (tag, len(log), len(gq))
        for k, (row, (err, rule)) in enumerate(zip(log, term)):
            mode, status, its, fac = row
            assert (mode, status) == (int(gq[k, 0]), int(gq[k, 1])), (tag, k)
            # (a run that ends by an acceptable-termination COUNTER -- 8 / 15 / 25 consecutive iterates within a level -- ends where the
            #  last digits put the first iterate of the streak: measured on sub-problem 1 of this scenario, 78 iterations in the
            #  oracle, 80 - 104 on the device depending on the front kernels' rounding (profiles/r04_ab_experiments.txt): half the
            #  count there, a fifth where the run ends at the tolerance)
            by_counter = rule >= 1 or int(gq[k, 6]) >= 1
            assert abs(its - gq[k, 2]) <= max(2, (0.5 if by_counter else 0.2) * gq[k, 2]), (tag, k, its, gq[k, 2])
            if rule == 3 or int(gq[k, 6]) == 3:
                report.append(f"{tag} sub-problem {k} (mode {mode}): device rule {rule} at scaled error {err:.2e}, oracle rule {int(gq[k, 6])} at {gq[k, 7]:.2e}")
        if tag == "quirks0_conv":
            assert rg["status"] == 0
            assert rel(rg["x"], G[f"{tag}_x"]) < TOL and abs(rg["obj_val"] - float(G[f"{tag}_obj_val"])) <= TOL * abs(float(G[f"{tag}_obj_val"])), tag
        else:
            # a truncated trajectory through sub-problems that are accurate to ~1e-6 along nearly flat directions: the decisions
            # above are exact, the point is held to the objective and reported
            dx = np.abs(rg["x"] - G[f"{tag}_x"])
            report.append(f"{tag}: objective {rg['obj_val']:.10e} against {float(G[f'{tag}_obj_val']):.10e}, |dx| max {dx.max():.2e}, "
                          f"median {np.median(dx):.2e}, 99th percentile {np.percentile(dx, 99):.2e}")
            assert abs(rg["obj_val"] - float(G[f"{tag}_obj_val"])) <= 1e-3 * abs(float(G[f"{tag}_obj_val"])), tag
            assert np.median(dx) < 1e-3, tag
    with capsys.disabled():
        print("\n[9241 fixture] sub-problems accepted by the third acceptable-termination rule (1e4 x ipm_tol = 1e-5):")
        for ln in report or ["  none"]:
            print("  " + ln)
