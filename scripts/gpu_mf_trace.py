"""Phase timing of k_mf_factor (zero / originals / extend-add / eliminate / write-back) per front, instance 0.
Builds a traced copy of the library (-DSQPHIP_MF_TRACE) into /tmp and runs one batched SQP iteration."""
import ctypes as C, os, subprocess, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from sqpsolver_jl_amd import _lib
so = os.environ.get("SQPHIP_TRACE_SO", "/tmp/libsqphip_trace.so")       # (a traced build made beforehand, or built here)
srcs = [os.path.join(_lib._CSRC, s) for s in _lib.SOURCES]
if "SQPHIP_TRACE_SO" not in os.environ:
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
ctx.sqp_reset(); ctx.sqp_run(1)
c = ctx.counters()
ns = c["n_supernodes"]
buf = np.zeros((ns, 8), dtype=np.int64)
L = _lib.lib()
L.sqphip_mf_trace_read.argtypes = [C.POINTER(C.c_longlong), C.c_int]
assert L.sqphip_mf_trace_read(buf.ctypes.data_as(C.POINTER(C.c_longlong)), ns) == 0
d = np.diff(buf[:, :6], axis=1) * 10e-3       # 100 MHz ticks -> microseconds
tot = d.sum(axis=1)
order = np.argsort(-tot)
names = ["zero", "orig", "extadd", "elim", "write"]
print(f"{case} B={B}: {ns} fronts; sum over fronts (us):", {n: round(float(v), 1) for n, v in zip(names, d.sum(axis=0))})
print("slowest fronts: idx total | " + " ".join(names))
for s in order[:12]:
    print(f"  {s:5d} {tot[s]:8.1f} | " + " ".join(f"{v:7.1f}" for v in d[s]))
# per size class (16-row tiles of the front): fronts, mean microseconds per phase
L.sqphip_kkt_symbolic  # (library loaded)
import collections
pos, st = pkg.kkt_symbolic(lay0.n, lay0.m, lay0.jrow, lay0.jcol, lay0.hrow, lay0.hcol, lay0.gL, lay0.gU)
print("symbolic:", {k: st[k] for k in ("n_supernodes", "n_levels", "max_front", "nnz_l")})
tot_by = collections.defaultdict(list)
for srow in range(ns):
    tot_by[int(round(tot[srow] // 5) * 5)].append(srow)
q = np.percentile(tot, [10, 50, 90, 99])
print("front time percentiles (us): p10 %.1f p50 %.1f p90 %.1f p99 %.1f" % tuple(q))
for lo, hi in ((0, 8), (8, 15), (15, 30), (30, 1e9)):
    sel = (tot >= lo) & (tot < hi)
    if sel.any():
        print(f"  fronts with total in [{lo},{hi}) us: {int(sel.sum()):4d}  mean phases " + " ".join(f"{n}={v:.1f}" for n, v in zip(names, d[sel].mean(axis=0))))
t0 = buf[:, 0].min()
print("span of the last factorisation (us):", (buf[:, 5].max() - t0) * 10e-3)

# shader-clock stamps inside the static front kernel (wave 0): tiles loaded | barrier | block 0: A | barrier | B | barrier | C ... | results
buf2 = np.zeros((ns, 16), dtype=np.int64)
L.sqphip_mf_trace2_read.argtypes = [C.POINTER(C.c_longlong), C.c_int]
if L.sqphip_mf_trace2_read(buf2.ctypes.data_as(C.POINTER(C.c_longlong)), ns) == 0:
    print("static front kernel, wave 0, cycles: load | bar | A(blk0) | bar | B(blk0) | bar | C(blk0)=to blk1 start | ... all blocks | bar | results")
    for srow in list(order[:8]) + list(order[60:64]):
        t = buf2[srow]
        if t[0] == 0: continue
        seg = [t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5], (t[9] - t[6]) if t[9] > 0 else -1, t[7] - t[2], t[8] - t[7], t[10] - t[8]]
        print(f"  front {srow:4d}: " + " ".join(f"{int(v):7d}" for v in seg))
        if t[11] > 0 and t[15] > 0:      # first 4 x 4 block of the diagonal tile: block in LDS -> read | LDL^T | rows x inverse | (issue) | update
            print("              first 4 x 4 block: start -> LDS written+sync " + str(int(t[11] - t[2])) + " | ten reads " + str(int(t[12] - t[11]))
                  + " | mf_ldl4 " + str(int(t[13] - t[12])) + " | apply4 + ic + product " + str(int(t[14] - t[13])) + " | mfma " + str(int(t[15] - t[14]))
                  + " | rest of the tile (three more blocks, records, tile out) " + str(int(t[3] - t[15])))

# the four-wave solve routines (thread 0): forward: loads issued | barrier | gather | 16-column blocks | rows below ; backward likewise
buf3 = np.zeros((ns, 16), dtype=np.int64)
L.sqphip_mf_trace3_read.argtypes = [C.POINTER(C.c_longlong), C.c_int]
if L.sqphip_mf_trace3_read(buf3.ctypes.data_as(C.POINTER(C.c_longlong)), ns) == 0:
    print("four-wave solves, cycles: fwd: wait+barrier | gather | blocks | rows-below   bwd: wait+barrier | L21'x | blocks   (nc, nr)")
    nc_nr = {}
    for srow in range(ns):
        t = buf3[srow]
        if t[9] == 0: continue
        f = [t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4]] if t[1] else [0] * 4
        b = [t[10] - t[9], t[11] - t[10], t[12] - t[11]]
        print(f"  front {srow:4d}: fwd " + " ".join(f"{int(v):6d}" for v in f) + "   bwd " + " ".join(f"{int(v):6d}" for v in b)
              + f"   fwd total {int(t[5] - t[1]) if t[1] else 0}  bwd total {int(t[12] - t[9])}  fwd-end -> bwd-start of same front {int(t[9] - t[5]) if t[1] else 0}")
    tt = buf3[buf3[:, 9] > 0]
    if len(tt):
        print("  span first fwd stamp -> last bwd stamp (cycles):", int(tt[:, 12].max() - tt[tt[:, 1] > 0][:, 1].min()) if (tt[:, 1] > 0).any() else -1)
    # streamed top solve: rows 0..63 backward-only launch, 64..127 forward + backward launch
    for base, name in ((0, "backward only"), (64, "forward + backward")):
        rows = buf3[base:base + 64] if ns >= base + 64 else np.zeros((0, 16), dtype=np.int64)
        rows = rows[rows[:, 0] > 0]
        if len(rows):
            print(f"k_mf_solve_top2 ({name}), cycles per step: compute | its wait at the barrier || loader: load | wait   (total {int(rows[-1, 2] - rows[0, 0])})")
            for j, t in enumerate(rows):
                fw = f"fwd: gather {int(t[8] - t[0]):6d} chain {int(t[9] - t[8]):6d}" if t[8] > t[0] else " " * 36
                bw = f"bwd: start {int(t[10] - max(t[0], t[9])):6d} L21 {int(t[11] - t[10]):6d} chain {int(t[12] - t[11]):6d} tail {int(t[1] - t[12]):6d}" if t[10] > t[0] else ""
                print(f"   step {j:2d}: {int(t[1] - t[0]):6d} {int(t[2] - t[1]):6d}  || {int(t[5] - t[4]):6d} {int(t[6] - t[5]):6d}    {fw} {bw}")
# vector stages of the last sweeps (instance 0, thread 0), shader cycles
vt = np.zeros(64, dtype=np.int64)
L.sqphip_vec_trace_read.argtypes = [C.POINTER(C.c_longlong)]
if L.sqphip_vec_trace_read(vt.ctypes.data_as(C.POINTER(C.c_longlong))) == 0:
    def seg(a, b): return int(vt[b] - vt[a]) if vt[a] and vt[b] else -1
    print("k_ipm_tail: refine accumulate", seg(0, 1), "| H sol, J sol", seg(1, 2), "| residual + reduce", seg(2, 3), "| next rhs", seg(3, 4))
    print("            step: (gap)", seg(4, 8), "expand", seg(8, 9), "| 2 reductions", seg(9, 10), "| update", seg(10, 11))
    print("            prepare: (gap)", seg(11, 16), "stage", seg(16, 17), "| H p", seg(17, 18), "| J p + barrier", seg(18, 19), "| residual loops", seg(19, 20), "| 6 reductions", seg(20, 21), "| rest", seg(21, 22))
    print("k_ipm_mid:  refine accumulate", seg(32, 33), "| H sol, J sol", seg(33, 34), "| residual + reduce", seg(34, 35), "| next rhs", seg(35, 36))
