"""Phase stamps of the spine kernel (k_mf_spine), instance 0, last factorisation: per front the microseconds spent in
finish_image | request of the next front | elimination + results | wait for the block / barrier | next image (zero + gather) | hand-over.
usage: gpu_spine_trace.py case118 B   (SQPHIP_TRACE_SO = a library built with -DSQPHIP_MF_TRACE beforehand, else built here)"""
import ctypes as C, os, subprocess, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from sqpsolver_jl_amd import _lib
so = os.environ.get("SQPHIP_TRACE_SO", "/tmp/libsqphip_trace.so")
if "SQPHIP_TRACE_SO" not in os.environ:
    srcs = [os.path.join(_lib._CSRC, s) for s in _lib.SOURCES]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-w",
                           "-DSQPHIP_MF_TRACE", "-mllvm", "-amdgpu-mfma-vgpr-form", "-o", so] + srcs + ["-ldl"])
_lib.SO_PATH = so
import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, contingency, CASES
case, B = sys.argv[1], int(sys.argv[2])
nb, ng, nl, seed = CASES[case]
base = acopf_synth(nb, ng, nl, seed); lay0 = acopf_layout(base)
ctx = pkg.Context(lay0.n, lay0.m, lay0.num_linear, lay0.jrow, lay0.jcol, lay0.hrow, lay0.hcol, lay0.xL, lay0.xU, lay0.gL,
                  lay0.gU, pkg.default_options(kkt_mode=2, tol_infeas=1e-6, tol_residual=1e-4, use_soc=1), batch=B)
ctx.acopf_attach(base, lay0)
for b in range(B):
    net = base if b == 0 else contingency(base, b, seed)
    ctx.acopf_set_instance(b, net, acopf_layout(net))
ctx.sqp_reset(); ctx.sqp_run(3)
buf = np.zeros((256, 16), dtype=np.int64)
L = _lib.lib()
L.sqphip_mf_trace3_read.argtypes = [C.POINTER(C.c_longlong), C.c_int]
assert L.sqphip_mf_trace3_read(buf.ctypes.data_as(C.POINTER(C.c_longlong)), 256) == 0
rows = buf[128:192, :7]
n = int((rows[:, 0] != 0).sum())
names = ["finish", "request", "elim+store", "wait", "next image", "hand-over"]
print(f"{case} B={B}: {n} spine fronts; microseconds per phase (instance 0, last factorisation)")
tot = np.zeros(6)
for k in range(n):
    d = np.diff(rows[k]) * 10e-3
    if k == n - 1: d[3:] = 0
    tot += d
    print(f"  front {k:2d}: " + " ".join(f"{nm} {v:6.1f}" for nm, v in zip(names, d)))
print("  sum     : " + " ".join(f"{nm} {v:6.1f}" for nm, v in zip(names, tot)), " total %.1f us" % ((rows[n - 1, 3] - rows[0, 0]) * 10e-3))
ctx.close()
