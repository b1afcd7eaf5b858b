"""on-box fp64 MFMA issue-rate probe of the library: python scripts/gpu_probe_peak.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd import _lib
v = C.c_double()
for _ in range(3):
    assert _lib.lib().sqphip_mfma_f64_peak(0, C.byref(v)) == 0
    print("fp64 MFMA probe: %.1f TFLOP/s" % v.value)
