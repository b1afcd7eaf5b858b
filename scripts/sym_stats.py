"""Host-only: statistics of the symbolic analysis (ordering, supernodes, fronts) for the synthetic case shapes.
Usage: python scripts/sym_stats.py [lib.so] case118 case1354 ..."""
import ctypes as C, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sqpsolver_jl_amd.acopf_synth import acopf_synth, acopf_layout, CASES

class Stats(C.Structure):
    _fields_ = [(k, C.c_int64) for k in ("order", "nnz_k_lower", "n_supernodes", "n_levels", "max_front", "max_cols",
                                         "nnz_l", "nnz_l_exact")] + [("flops", C.c_double), ("flops_exact", C.c_double),
                                                                     ("front_doubles", C.c_int64)]
args = sys.argv[1:]
so = args.pop(0) if args and args[0].endswith(".so") else None
if so is None:
    from sqpsolver_jl_amd import _lib
    so = _lib.SO_PATH
L = C.CDLL(so)
lp, dp, ip = C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int32)
L.sqphip_kkt_symbolic.argtypes = [C.c_int64, C.c_int64, C.c_int64, lp, lp, C.c_int64, lp, lp, dp, dp, C.c_int32,
                                  C.c_int32, C.c_int32, C.c_double, ip, C.POINTER(Stats)]
def P(a, t): return a.ctypes.data_as(t)
for case in args or ["case14", "case118", "case1354"]:
    nb, ng, nl, seed = CASES[case]
    lay = acopf_layout(acopf_synth(nb, ng, nl, seed))
    for cond in (1, 0):
        for rav in (1, 0):
            for sf, zf in ((32, 0.25), (64, 0.25), (16, 0.1), (1, 0.0)):
                st = Stats()
                t0 = time.time()
                rc = L.sqphip_kkt_symbolic(lay.n, lay.m, len(lay.jrow), P(lay.jrow, lp), P(lay.jcol, lp), len(lay.hrow),
                                           P(lay.hrow, lp), P(lay.hcol, lp), P(lay.gL, dp), P(lay.gU, dp), cond, rav, sf, zf,
                                           None, C.byref(st))
                dt = time.time() - t0
                assert rc == 0
                print(f"{case} cond={cond} rows_after={rav} small={sf} zf={zf}: order {st.order} nnzK {st.nnz_k_lower} "
                      f"sn {st.n_supernodes} lev {st.n_levels} maxfront {st.max_front} maxcols {st.max_cols} "
                      f"nnzL {st.nnz_l} (exact {st.nnz_l_exact}) MF {st.flops/1e6:.1f} (exact {st.flops_exact/1e6:.1f}) "
                      f"front MB {st.front_doubles*8/1e6:.1f}  [{dt:.2f}s]")
