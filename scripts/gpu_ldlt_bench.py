"""LDL^T micro-benchmark only (for rocprofv3): python scripts/gpu_ldlt_bench.py N B reps"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd import _lib
L = _lib.lib()
N, B, reps = (int(v) for v in sys.argv[1:4])
s = C.c_double(); st = C.c_double(); nl = C.c_int64()
assert L.sqphip_ldlt_bench(0, B, N, reps, C.byref(s), C.byref(st), C.byref(nl)) == 0
fl = B * N**3 / 3
print(f"N={N} B={B}: {s.value*1e3:.3f} ms/factor-batch = {fl/s.value/1e12:.2f} TFLOP/s; trailing events {st.value*1e3:.3f} ms over {nl.value} launches")
