"""Look-ahead vs single-stream factorisation, bitwise: python scripts/gpu_ldlt_stress.py N B reps"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd import _lib
L = _lib.lib()
N, B, reps = (int(v) for v in sys.argv[1:4])
L.sqphip_ldlt_stress.argtypes = [C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.POINTER(C.c_int32)]
m = C.c_int32()
assert L.sqphip_ldlt_stress(0, B, N, reps, C.byref(m)) == 0
print(f"N={N} B={B}: {m.value} of {reps} repetitions differ")
