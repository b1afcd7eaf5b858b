"""GPU sanity: LDL^T factor/solve parity against the CPU oracle + micro-benchmark."""
import ctypes as C, sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd import _lib
from oracle import oracle as O

L = _lib.lib()
dp = C.POINTER(C.c_double)
def d(a): return a.ctypes.data_as(dp)

def qd(N, n1, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((N, N)) * 0.3
    A = (A + A.T) / 2
    dg = np.concatenate([np.full(n1, 1.0), np.full(N - n1, -1.0)]) * (0.3 * np.sqrt(N) * 3 + rng.uniform(0.5, 1.5, N))
    A[np.diag_indices(N)] = dg
    return A

out = {}
for N, B in ((6, 2), (64, 2), (100, 3), (307, 4), (600, 2), (307, 8), (700, 16)):
    As = np.stack([qd(N, N * 2 // 5, 10 + b) for b in range(B)])
    rhs = np.random.default_rng(5).standard_normal((B, N))
    Af = np.ascontiguousarray(np.stack([np.asfortranarray(a).ravel(order="F") for a in As]))
    dinv = np.zeros((B, N)); npos = np.zeros(B, dtype=np.int32)
    A_dev = Af.copy()
    rc = L.sqphip_ldlt_factor_host(0, B, N, d(A_dev), d(dinv), npos.ctypes.data_as(C.POINTER(C.c_int32)))
    assert rc == 0
    x = rhs.copy()
    rc = L.sqphip_ldlt_solve_host(0, B, N, d(Af), d(x)); assert rc == 0
    errs = []
    for b in range(B):
        a_o, dinv_o, np_o, _ = O.ldlt_factor(As[b], N)
        Lg = np.tril(A_dev[b].reshape(N, N, order="F"), -1); Lo = np.tril(a_o, -1)
        eL = np.abs(Lg - Lo).max() / max(1, np.abs(Lo).max())
        eD = np.abs(dinv[b] - dinv_o).max() / np.abs(dinv_o).max()
        xo = np.linalg.solve(As[b], rhs[b])
        eX = np.abs(x[b] - xo).max() / np.abs(xo).max()
        errs.append((eL, eD, eX, int(npos[b]), int(np_o)))
    print("N", N, "B", B, "errs(L,D,x,npos,npos_oracle)", errs, flush=True)
    out[f"N{N}"] = errs
for N, B, reps in ((307, 64, 5), (2813, 8, 2), (2813, 64, 2)):
    s = C.c_double(); st = C.c_double(); nl = C.c_int64()
    rc = L.sqphip_ldlt_bench(0, B, N, reps, C.byref(s), C.byref(st), C.byref(nl)); assert rc == 0
    fl = B * N**3 / 3
    print(f"bench N={N} B={B}: {s.value*1e3:.3f} ms/factor-batch  {fl/s.value/1e12:.2f} TFLOP/s total; trailing {st.value*1e3:.3f} ms ({nl.value} launches) -> {fl/max(st.value,1e-12)/1e12:.2f} TFLOP/s if all flops there", flush=True)
    out[f"bench_N{N}_B{B}"] = dict(sec=s.value, sec_trailing=st.value, launches=nl.value, tflops=fl/s.value/1e12)
json.dump(out, open("gpurun_out/ldlt_check.json", "w"), indent=1, default=float)
