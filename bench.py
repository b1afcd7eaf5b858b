#!/usr/bin/env python3
"""Headline benchmark: QP sub-problems per second of the batched SQP-TR hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one SQP-TR outer iteration of every instance of the rank's batch: device ACOPF
evaluation, the trust-region QP (or feasibility-restoration / second-order-correction) sub-problem
solved by the on-device interior-point method, merit + ratio test.  Workload at N = 1: the per-GPU
shard of BASELINE.json configs[3] -- 64 IEEE-118-shaped ACOPF contingency scenarios (512 over 8 GPUs,
weak scaling: 64 per rank), KKT order 2813 condensed to 2069 (options.kkt_condense) and ordered into 23 independent
leading tiles + a dense remainder of 673 (options.kkt_tile_order), fp64 LDL^T, synthetic
data of that shape, SQP options of
/root/reference/examples/acopf/opf.jl:76-79.  Inputs are resident in HBM before the timed region.
Ranks never exchange iterates; the timed region ends with one all-gather of (ret, iter, done) per
instance over RCCL.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

_ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X datasheet fp64 matrix peak; MI355X_MICROARCH.md has no fp64 row.  The on-box
                               # register-resident MFMA probe (printed next to it) sustains 77.6 of it.


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="case118", choices=["case14", "case118"])
    ap.add_argument("--batch", type=int, default=64, help="instances per GPU")
    ap.add_argument("--groups", type=int, default=1,
                    help="instance groups per GPU, each a context with its own HIP stream pair, driven "
                         "concurrently so that the latency-bound panel / solve phases of one group "
                         "overlap the MFMA-bound updates of another (2 gives about +5 % QP/s on MI355X but the "
                         "HIP-event kernel timing then includes queueing behind the other group, so the "
                         "default keeps one group and a clean roofline measurement)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--literal-quirks", type=int, default=1)
    ap.add_argument("--kkt-tile-order", type=int, default=None,
                    help="options.kkt_tile_order (default: the library default)")
    ap.add_argument("--kkt-condense", type=int, default=None,
                    help="options.kkt_condense (default: the library default)")
    ap.add_argument("--ipm-corrector", type=int, default=1,
                    help="options.ipm_corrector (library default 1: predictor-corrector interior-point iterations)")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the HIP-event timing of k_trailing")
    ap.add_argument("--sqp-options", default="example", choices=["example", "defaults"],
                    help="example: tol_infeas 1e-6, tol_residual 1e-4, use_soc (examples/acopf/opf.jl:76-79, the headline); "
                         "defaults: the reference's Parameters defaults (parameters.jl:17-29: tol_residual 1e-6, no SOC)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the multi-rank flow "
                         "on a box with fewer GPUs than ranks, together with --one-device)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")

    import sqpsolver_jl_amd as pkg
    from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, contingency, CASES
    from sqpsolver_jl_amd.shard import shard_range, gather_status
    from sqpsolver_jl_amd import _lib

    nb, ng, nl, seed = CASES[args.workload]
    B = args.batch
    total = B * world
    lo, hi = shard_range(total, world, rank)
    base = acopf_synth(nb, ng, nl, seed)
    lay0 = acopf_layout(base)
    # examples/acopf/opf.jl:72-80
    use_soc = 1 if args.sqp_options == "example" else 0
    sqp_kw = dict(tol_infeas=1e-6, tol_residual=1e-4, use_soc=1) if use_soc else {}
    lin_kw = {} if args.kkt_condense is None else {"kkt_condense": args.kkt_condense}
    if args.kkt_tile_order is not None:
        lin_kw["kkt_tile_order"] = args.kkt_tile_order
    opts = pkg.default_options(max_iter=3000, literal_quirks=args.literal_quirks, device=local_rank,
                               ipm_corrector=args.ipm_corrector, **lin_kw, **sqp_kw)
    import threading
    G = max(1, min(args.groups, hi - lo))
    ctxs, nets = [], []
    for gi in range(G):
        glo, ghi = shard_range(hi - lo, G, gi)
        ctx = pkg.Context(lay0.n, lay0.m, lay0.num_linear, lay0.jrow, lay0.jcol, lay0.hrow, lay0.hcol,
                          lay0.xL, lay0.xU, lay0.gL, lay0.gU, opts, batch=ghi - glo)
        ctx.acopf_attach(base, lay0)
        for b, s_id in enumerate(range(lo + glo, lo + ghi)):
            net = base if s_id == 0 else contingency(base, s_id, seed)
            lay = acopf_layout(net)
            ctx.acopf_set_instance(b, net, lay)
            nets.append((net, lay))
        ctx.sqp_reset()
        ctxs.append(ctx)

    def counters_sum():
        tot = {}
        for c in ctxs:
            for k, v in c.counters().items():
                tot[k] = tot.get(k, 0) + v
        return tot

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    dev = torch.device("cuda", local_rank) if args.backend == "nccl" else torch.device("cpu")   # collective buffers

    def run_steps(k):
        """k outer SQP-TR iterations of every instance of the shard (continuous batching inside the
        library: an instance never waits for the slowest sub-problem of the batch), then the status
        all-gather across ranks."""
        ths = [threading.Thread(target=c.sqp_run, args=(k,)) for c in ctxs]   # ctypes drops the GIL
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        parts = [c.sqp_status() for c in ctxs]
        ret, it, done = (np.concatenate([p[i] for p in parts]) for i in range(3))
        return gather_status(ret, it, done, total, device=dev)

    if args.warmup > 0:
        run_steps(args.warmup)
    c0 = counters_sum()
    for c in ctxs:
        c.set_timing(not args.no_kernel_timing)
    sync()
    t0 = time.perf_counter()
    g_ret, g_it, g_done = run_steps(args.steps)
    sync()
    t1 = time.perf_counter()
    for c in ctxs:
        c.set_timing(False)
    c1 = counters_sum()

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    stats = torch.tensor([c1["n_qp"] - c0["n_qp"], c1["n_ipm_iter"] - c0["n_ipm_iter"],
                          c1["n_factor"] - c0["n_factor"]], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    elapsed = float(elapsed.item())
    n_qp, n_ipm, n_fac = (float(v) for v in stats.tolist())

    # roofline of the dominant kernel (k_trailing, fp64 MFMA), rank-0 local measurement
    cinfo = ctxs[0].counters()
    N = int(cinfo["kkt_order"])         # order of the factorised matrices (n + m, or the condensed order)
    N_full = lay0.n + lay0.m
    N_c = lay0.n + int(np.sum(lay0.gL == lay0.gU))       # condensed order (before tile padding)
    lead = int(cinfo["lead_tiles"])
    loc_fac = c1["n_factor"] - c0["n_factor"]
    tr_sec = c1["trailing_seconds"] - c0["trailing_seconds"]
    tr_launch = c1["trailing_launches"] - c0["trailing_launches"]
    # algorithmic flops of the k_trailing launches of one factorisation, as the library schedules them
    achieved = loc_fac * float(cinfo["trailing_flops_per_factor"]) / tr_sec / 1e12 if tr_sec > 0 else 0.0
    probe = None
    if rank == 0:
        import ctypes as C
        v = C.c_double()
        if _lib.lib().sqphip_mfma_f64_peak(local_rank, C.byref(v)) == 0:
            probe = v.value
    traffic = None
    tpath = os.path.join(_ROOT, "profiles", "trailing_traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU restatement (NOT Julia/Ipopt) on a bounded sample: instance 0, first 6 outer iterations (about 12 s)
        from oracle import oracle as O
        try:
            cores = max(1, min(16, len(os.sched_getaffinity(0))))   # the GPU box grants 16 host cores per GPU
        except AttributeError:
            cores = max(1, min(16, os.cpu_count() or 1))
        oo = O.default_options(max_iter=6, literal_quirks=args.literal_quirks, num_threads=cores,
                               ipm_corrector=args.ipm_corrector, kkt_condense=int(opts.kkt_condense),
                               kkt_tile_order=int(opts.kkt_tile_order), **sqp_kw)
        ro = O.sqp_solve(O.problem_acopf(*nets[0]), oo)
        N_cpu = lay0.n + (int(np.sum(lay0.gL == lay0.gU)) if int(opts.kkt_condense) else lay0.m)
        cpu = {"value": ro["n_qp"] / ro["qp_seconds"] if ro["qp_seconds"] > 0 else 0.0,
               "unit": "QP subproblems/s", "cores": cores, "kind": "port",
               "sample": f"{args.workload} scenario 0, first 6 SQP-TR iterations = {ro['n_qp']} sub-problems, "
                         f"{ro['n_factor']} dense LDL^T of order {N_cpu} (plain dense, in the product's order), "
                         f"{ro['qp_seconds']:.1f} s; CPU restatement "
                         f"(oracle/), not Julia/Ipopt"}

    if rank == 0:
        out = {
            "metric": "QP subproblems/sec on batched ACOPF; fp64 KKT LDL^T TFLOPS vs MFMA peak",
            "value": n_qp / elapsed if elapsed > 0 else 0.0,
            "unit": "QP subproblems/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / max(1, args.steps),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{B} x IEEE-{nb}-bus-shaped ACOPF contingency scenarios per GPU "
                                   f"(BASELINE.json configs[3] shard), KKT order {N_full}"
                                   + (f" condensed to {N_c}" if int(opts.kkt_condense) else "")
                                   + (f", ordered into {lead} independent leading tiles + a dense remainder of {N - 64 * lead}"
                                      if lead else "") + ", fp64 LDL^T, SQP-TR outer iterations",
                       "instances_total": total, "kkt_order": N, "use_soc": use_soc, "sqp_options": args.sqp_options,
                       "groups_per_gpu": G,
                       "literal_quirks": args.literal_quirks, "ipm_corrector": args.ipm_corrector,
                       "kkt_condense": int(opts.kkt_condense), "kkt_order_full": N_full,
                       "kkt_tile_order": int(opts.kkt_tile_order), "independent_lead_tiles": int(cinfo["lead_tiles"]),
                       "qp_solved": n_qp, "ipm_iterations": n_ipm, "kkt_factorisations": n_fac,
                       "ldlt_dense_equivalent_tflops_wall": n_fac * ((N_c if int(opts.kkt_condense) else N_full) ** 3 / 3.0) / elapsed / 1e12 if elapsed > 0 else 0.0,
                       "instances_done": int(np.sum(g_done))},
            "roofline": {"bound": "mfma", "kernel": "k_trailing + k_trailing_list (v_mfma_f64_16x16x4_f64)",
                         "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "launches": tr_launch, "avg_launch_ms": 1e3 * tr_sec / tr_launch if tr_launch else None,
                         "onbox_mfma_probe_tflops": probe},
            "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    for c in ctxs:
        c.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
