"""ctypes loader of libsqphip.so (the HIP extension).  Fails loudly when the library is absent:
there is no CPU fallback anywhere in this package."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SO_PATH = os.path.join(_CSRC, "libsqphip.so")
if os.environ.get("SQPHIP_SO"):        # experiment aid: another build of the library (A / B runs on one box); never set in normal use
    SO_PATH = os.path.abspath(os.environ["SQPHIP_SO"])
SOURCES = ["ldlt.hip", "kernel_api.hip", "ipm.hip", "acopf.hip", "sqp.hip", "api.hip", "order.hip", "symbolic.hip",
           "mfplan.hip", "mfront.hip", "comm.hip"]
HEADERS = ["sqphip_internal.hpp", "ctx.hpp", "sparse.hpp", "dev_util.hpp", "mf_dev.hpp", "acopf_dev.hpp", os.path.join("..", "..", "include", "sqphip.h"),
           os.path.join("..", "..", "include", "sqphip_test_hooks.h")]

_lib = None


# per-file extra flags.  mfront.hip: keep the MFMA accumulators of the front kernels in VGPRs -- the elimination works on
# the accumulator tiles with ordinary vector instructions between the MFMAs, and with the accumulators homed in AGPRs
# the compiler copied all of them (72 v_accvgpr_read per four-column step) in and out at every step
# mfront.hip: MFMA accumulators in VGPRs; and a pragma-unroll budget that covers the static front kernels of nine to twelve tile
# rows (at the default budget some of their tile loops stay rolled and the tiles they index move to scratch: 416 - 544 B per lane)
EXTRA_FLAGS = {"mfront.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form", "-mllvm", "-pragma-unroll-threshold=131072"]}
_OBJ = os.path.join(_CSRC, "build")


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950: every translation unit to an object (only the stale ones, in parallel), then linked
    into csrc/libsqphip.so (in-tree)."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(_CSRC, s))]
    hdrs = [os.path.join(_CSRC, h) for h in HEADERS if os.path.exists(os.path.join(_CSRC, h))]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(_OBJ, exist_ok=True)
    hdr_t = max([os.path.getmtime(h) for h in hdrs] + [os.path.getmtime(os.path.abspath(__file__))])
    jobs = []
    for sname in srcs:
        src, obj = os.path.join(_CSRC, sname), os.path.join(_OBJ, sname + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append((sname, src, obj))

    def compile_one(job):
        sname, src, obj = job
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-Wno-unused-value", "-Wno-pass-failed",
               "-Rpass-analysis=kernel-resource-usage"] + EXTRA_FLAGS.get(sname, []) + ["-o", obj, src]
        if verbose:
            print(" ".join(cmd))
        proc = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        with open(obj + ".remarks", "w") as fh:
            fh.write(proc.stderr if proc.returncode == 0 else "")
        return sname, proc.returncode, proc.stderr, cmd

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            results = list(ex.map(compile_one, jobs))
        for sname, rc, err, cmd in results:
            other = [ln for ln in err.splitlines() if "-Rpass-analysis=kernel-resource-usage" not in ln]
            if rc != 0 or verbose:
                sys.stderr.write("\n".join(other) + "\n")
            if rc != 0:
                raise subprocess.CalledProcessError(rc, cmd)
    objs = [os.path.join(_OBJ, sname + ".o") for sname in srcs]
    if jobs or not os.path.exists(SO_PATH) or any(os.path.getmtime(o) > os.path.getmtime(SO_PATH) for o in objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO_PATH] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        remarks = ""
        for o in objs:
            if os.path.exists(o + ".remarks"):
                remarks += open(o + ".remarks").read()
        _write_kernel_resources(remarks)
    return SO_PATH


RESOURCES_PATH = os.path.join(_CSRC, "kernel_resources.json")


def _write_kernel_resources(remarks: str) -> None:
    """Registers, scratch and LDS of every kernel as the compiler reports them (tests/test_abi.py guards the scratch: a
    kernel that stops being inlined in one piece takes its 1 KB argument struct through scratch memory)."""
    import json
    import re
    out, cur = {}, None
    for ln in remarks.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\d+)", ln)
        if m and cur is not None:
            cur[m.group(1).split(" [")[0]] = int(m.group(2))
    with open(RESOURCES_PATH, "w") as fh:
        json.dump(out, fh, indent=0, sort_keys=True)


class Options(C.Structure):
    _fields_ = [(k, C.c_double) for k in
                ("tol_direction", "tol_residual", "tol_infeas", "init_mu", "max_mu", "tr_size",
                 "rho", "eta", "tau", "min_alpha")] + \
               [("max_iter", C.c_int32), ("use_soc", C.c_int32), ("literal_quirks", C.c_int32),
                ("ipm_tol", C.c_double), ("ipm_max_iter", C.c_int32), ("ipm_phase1", C.c_int32),
                ("device", C.c_int32), ("ipm_corrector", C.c_int32),
                ("kkt_condense", C.c_int32), ("kkt_tile_order", C.c_int32), ("kkt_mode", C.c_int32),
                ("ipm_warm_start", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [("n_qp", C.c_int64), ("n_ipm_iter", C.c_int64), ("n_factor", C.c_int64),
                ("ldlt_flops", C.c_double), ("ldlt_seconds", C.c_double),
                ("trailing_seconds", C.c_double), ("solve_seconds", C.c_double),
                ("total_seconds", C.c_double), ("trailing_launches", C.c_int64), ("kkt_order", C.c_int64),
                ("lead_tiles", C.c_int64), ("trailing_flops_per_factor", C.c_double),
                ("sparse", C.c_int64), ("nnz_k", C.c_int64), ("nnz_l", C.c_int64), ("n_supernodes", C.c_int64),
                ("n_levels", C.c_int64), ("max_front", C.c_int64), ("factor_flops", C.c_double),
                ("front_doubles", C.c_int64), ("cb_doubles", C.c_int64), ("factor_launches", C.c_int64),
                ("solve_launches", C.c_int64), ("n_sweeps", C.c_int64), ("n_solve", C.c_int64), ("n_groups", C.c_int64),
                ("nnz_l_top", C.c_int64), ("nnz_k_top", C.c_int64), ("cols_top", C.c_int64)]


class SymbolicStats(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("order", "nnz_k_lower", "n_supernodes", "n_levels", "max_front", "max_cols",
                                         "nnz_l", "nnz_l_exact")] + \
               [("flops", C.c_double), ("flops_exact", C.c_double), ("front_doubles", C.c_int64)]


def lib():
    """Load libsqphip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                f"{SO_PATH} is missing: build it with __graft_entry__.build() "
                "(hipcc --offload-arch=gfx950); this package has no CPU fallback")
        L = C.CDLL(SO_PATH)
        dp, ip, lp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
        vp = C.c_void_p
        if hasattr(L, "sqphip_default_options"):
            L.sqphip_default_options.argtypes = [C.POINTER(Options)]
        L.sqphip_ldlt_factor_host.argtypes = [C.c_int32, C.c_int32, C.c_int64, dp, dp, ip]
        L.sqphip_ldlt_solve_host.argtypes = [C.c_int32, C.c_int32, C.c_int64, dp, dp]
        L.sqphip_ldlt_bench.argtypes = [C.c_int32, C.c_int32, C.c_int64, C.c_int32, dp, dp, lp]
        L.sqphip_mfma_f64_peak.argtypes = [C.c_int32, dp]
        if hasattr(L, "sqphip_create"):
            L.sqphip_create.argtypes = [C.POINTER(vp), C.c_int64, C.c_int64, C.c_int64,
                                        C.c_int64, lp, lp, C.c_int64, lp, lp, dp, dp, dp, dp,
                                        C.POINTER(Options), C.c_int32]
            L.sqphip_destroy.argtypes = [vp]
            L.sqphip_last_error.restype = C.c_char_p
            L.sqphip_last_error.argtypes = [vp]
            L.sqphip_set_bounds.argtypes = [vp, C.c_int32, dp, dp, dp, dp]
            L.sqphip_qp_solve.argtypes = [vp, C.c_int32, dp, C.c_double, C.c_double, dp, dp, dp, dp,
                                          dp, dp, dp, dp, dp, ip]
            L.sqphip_qp_stats.argtypes = [vp, ip, ip]
            L.sqphip_qp_termination.argtypes = [vp, ip, dp]
            L.sqphip_sqp_qp_log_term.argtypes = [vp, C.c_int32, dp, ip, C.c_int32, ip]
            L.sqphip_get_termination_counters.argtypes = [vp, C.POINTER(C.c_int64)]
            L.sqphip_norm_violations.argtypes = [vp, dp, dp, C.c_int32, dp]
            L.sqphip_kt_residuals.argtypes = [vp, dp, dp, dp, dp, dp, dp]
            L.sqphip_norm_complementarity.argtypes = [vp, dp, dp, C.c_int32, dp]
            L.sqphip_compute_phi.argtypes = [vp, C.c_double, dp, dp, C.c_double, C.c_int32, dp]
            L.sqphip_compute_qmodel.argtypes = [vp, dp, dp, dp, dp, dp, dp, C.c_double, C.c_int32, dp]
            L.sqphip_compute_derivative.argtypes = [vp, dp, dp, dp, C.c_double, dp]
            L.sqphip_compute_derivative_full.argtypes = [vp, dp, dp, dp, C.c_double, dp, C.c_int32, dp, dp]
            L.sqphip_compute_mu_rule_dev.argtypes = [vp, C.c_int32, C.c_int64, C.c_double, dp, dp, dp, dp, dp, dp, dp]
            L.sqphip_acopf_armijo.argtypes = [vp, C.c_int32, dp, dp] + [C.c_double] * 6 + [C.c_int32, dp, ip, ip]
            L.sqphip_tr_update.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double,
                                           C.c_double, C.c_double, ip, dp]
            L.sqphip_acopf_attach.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, ip, ip, ip, ip, ip,
                                              ip, dp, C.c_int32]
            L.sqphip_acopf_attach_acr.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, ip, ip, ip, ip, ip,
                                              ip, dp, C.c_int32]
            L.sqphip_acopf_attach_acwr.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, ip, ip, ip, ip, ip, ip, dp, C.c_int32,
                                                   C.c_int32, ip, ip, ip, dp, dp, dp]
            L.sqphip_acopf_set_instance.argtypes = [vp, C.c_int32, dp, dp, dp, dp]
            L.sqphip_dense_attach.argtypes = [vp, dp, dp, C.c_double]
            L.sqphip_dense_set_instance.argtypes = [vp, C.c_int32, dp, dp]
            L.sqphip_acopf_set_shunts.argtypes = [vp, C.c_int32, ip, dp, dp]
            L.sqphip_acopf_set_dclines.argtypes = [vp, C.c_int32, dp]
            L.sqphip_kkt_order.argtypes = [C.c_int64, C.c_int64, C.c_int64, lp, lp, C.c_int64, lp, lp, dp, dp, C.c_int32,
                                           ip, ip, ip]
            L.sqphip_kkt_symbolic.argtypes = [C.c_int64, C.c_int64, C.c_int64, lp, lp, C.c_int64, lp, lp, dp, dp, C.c_int32,
                                              C.c_int32, C.c_int32, C.c_double, ip, C.POINTER(SymbolicStats)]
            L.sqphip_mf_host_solve.argtypes = [C.c_int64, C.c_int64, C.c_int64, lp, lp, C.c_int64, lp, lp, dp, dp, C.c_int32,
                                               dp, dp, dp, dp, dp, ip, C.c_double, C.c_double, dp, dp, dp, ip]
            L.sqphip_mf_host_top2_err.argtypes = []
            L.sqphip_mf_host_top2_err.restype = C.c_double
            L.sqphip_mf_host_spine_err.argtypes = []
            L.sqphip_mf_host_spine_err.restype = C.c_double
            L.sqphip_mf_solve_test.argtypes = [vp, C.c_int32, dp, dp, dp, dp, dp, ip, C.c_double, C.c_double, dp, dp, dp, dp]
            L.sqphip_acopf_eval.argtypes = [vp, C.c_int32, dp, C.c_double, dp, dp, dp, dp, dp, dp]
            L.sqphip_sqp_reset.argtypes = [vp]
            L.sqphip_sqp_run.argtypes = [vp, C.c_int32]
            L.sqphip_sqp_get.argtypes = [vp, C.c_int32, dp, dp, dp, dp, dp, dp, ip, ip]
            L.sqphip_sqp_status.argtypes = [vp, ip, ip, ip]
            L.sqphip_sqp_trace.argtypes = [vp, C.c_int32, dp, C.c_int32, ip]
            L.sqphip_comm_unique_id.argtypes = [vp]
            L.sqphip_comm_init.argtypes = [vp, vp, C.c_int32, C.c_int32]
            L.sqphip_gather_status.argtypes = [vp, C.c_int32, ip, ip, ip]
            L.sqphip_comm_destroy.argtypes = [vp]
            L.sqphip_get_counters.argtypes = [vp, C.POINTER(Counters)]
            L.sqphip_get_mode_counters.argtypes = [vp, C.POINTER(C.c_int64)]
            L.sqphip_sqp_qp_log.argtypes = [vp, C.c_int32, ip, C.c_int32, ip]
            L.sqphip_sqp_last_request.argtypes = [vp, C.c_int32, ip, dp, dp, dp, dp, dp, dp, dp]
            L.sqphip_sqp_stream_begin.argtypes = [vp, C.c_int32]
            L.sqphip_sqp_stream_set.argtypes = [vp, C.c_int32] + [dp] * 8
            L.sqphip_sqp_stream_run.argtypes = [vp]
            L.sqphip_sqp_stream_get.argtypes = [vp, C.c_int32, dp, dp, ip, ip]
            L.sqphip_sqp_work.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
            L.sqphip_reset_counters.argtypes = [vp]
            L.sqphip_set_timing.argtypes = [vp, C.c_int32]
            L.sqphip_get_kernel_times.argtypes = [vp, dp, C.POINTER(C.c_int64), C.c_int32]
        _lib = L
    return _lib


EXPORTS = [
    "sqphip_default_options", "sqphip_create", "sqphip_destroy", "sqphip_last_error",
    "sqphip_set_bounds", "sqphip_qp_solve", "sqphip_qp_stats", "sqphip_qp_termination", "sqphip_sqp_qp_log_term", "sqphip_get_termination_counters", "sqphip_norm_violations",
    "sqphip_kt_residuals", "sqphip_norm_complementarity", "sqphip_compute_phi",
    "sqphip_compute_qmodel", "sqphip_compute_derivative", "sqphip_compute_derivative_full", "sqphip_compute_mu_rule_dev",
    "sqphip_acopf_armijo", "sqphip_tr_update",
    "sqphip_kkt_order", "sqphip_kkt_symbolic", "sqphip_mf_host_solve", "sqphip_mf_host_top2_err", "sqphip_mf_host_spine_err", "sqphip_mf_solve_test", "sqphip_acopf_attach", "sqphip_acopf_attach_acr", "sqphip_acopf_attach_acwr", "sqphip_acopf_set_shunts", "sqphip_acopf_set_dclines", "sqphip_acopf_set_instance", "sqphip_dense_attach", "sqphip_dense_set_instance", "sqphip_acopf_eval", "sqphip_sqp_reset",
    "sqphip_sqp_run", "sqphip_sqp_get", "sqphip_sqp_status", "sqphip_sqp_trace",
    "sqphip_comm_available", "sqphip_comm_unique_id", "sqphip_comm_init", "sqphip_gather_status", "sqphip_comm_destroy",
    "sqphip_get_counters", "sqphip_get_mode_counters", "sqphip_sqp_work", "sqphip_sqp_stream_begin", "sqphip_sqp_stream_set", "sqphip_sqp_stream_run", "sqphip_sqp_stream_get", "sqphip_sqp_stream_assign", "sqphip_sqp_stream_append", "sqphip_sqp_stream_release", "sqphip_sqp_stream_run_some", "sqphip_sqp_last_request", "sqphip_sqp_qp_log", "sqphip_reset_counters", "sqphip_set_timing", "sqphip_get_kernel_times", "sqphip_ldlt_factor_host",
    "sqphip_ldlt_solve_host", "sqphip_ldlt_bench", "sqphip_ldlt_stress", "sqphip_mfma_f64_peak", "sqphip_armijo_alpha", "sqphip_compute_mu_rule",
]
