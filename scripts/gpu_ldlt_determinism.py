"""Factor + solve the same random quasi-definite batch repeatedly; compare bitwise: python scripts/gpu_ldlt_determinism.py N B reps"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd import _lib
L = _lib.lib()
N, B, reps = (int(v) for v in sys.argv[1:4])
rng = np.random.default_rng(5)
n1 = N * 2 // 5
A0 = np.zeros((B, N, N))
for b in range(B):
    M = rng.uniform(-0.05, 0.05, (N, N)); M = np.tril(M) + np.tril(M, -1).T
    d = np.concatenate([np.full(n1, 0.05 * N + 1.0), -np.full(N - n1, 0.05 * N + 1.0)]) + rng.uniform(-0.5, 0.5, N)
    M[np.arange(N), np.arange(N)] = d
    A0[b] = M
rhs0 = rng.uniform(-1, 1, (B, N))
ref = None
for rep in range(reps):
    A = np.ascontiguousarray(A0.transpose(0, 2, 1)).copy()     # column-major per instance
    dinv = np.zeros((B, N)); npos = np.zeros(B, dtype=np.int32)
    assert L.sqphip_ldlt_factor_host(0, B, N, A.ctypes.data_as(C.POINTER(C.c_double)), dinv.ctypes.data_as(C.POINTER(C.c_double)), npos.ctypes.data_as(C.POINTER(C.c_int32))) == 0
    A2 = np.ascontiguousarray(A0.transpose(0, 2, 1)).copy(); x = rhs0.copy()
    assert L.sqphip_ldlt_solve_host(0, B, N, A2.ctypes.data_as(C.POINTER(C.c_double)), x.ctypes.data_as(C.POINTER(C.c_double))) == 0
    low = np.tril_indices(N, -1)
    cur = (np.stack([A[b].T[low] for b in range(B)]), dinv.copy(), x.copy())
    if ref is None:
        ref = cur
    else:
        dl = np.max(np.abs(cur[0] - ref[0])); dd = np.max(np.abs(cur[1] - ref[1])); dx = np.max(np.abs(cur[2] - ref[2]))
        print(f"rep {rep}: max|dL| {dl:.1e} max|d dinv| {dd:.1e} max|dx| {dx:.1e}", "SAME" if dl == dd == dx == 0 else "DIFFERENT")
