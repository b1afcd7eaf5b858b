"""Interior-point iteration counts per outer iteration, device against oracle, on the convergent 1354-bus shape
(the data behind the allowance of tests/test_gpu_parity_depth.py::test_case1354_geo_converges_like_the_oracle)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import sqpsolver_jl_amd as pkg
from sqpsolver_jl_amd.acopf_synth import acopf_layout, contingency, synth_case, CASES
from oracle import oracle as O
from test_gpu_parity import _run_batch
base = synth_case("case1354")
nets = [base, contingency(base, 17, CASES["case1354"][3])]
lays = [acopf_layout(nt) for nt in nets]
kw = dict(max_iter=40, tol_infeas=1e-6, tol_residual=1e-4, use_soc=1, literal_quirks=0)
ctx = _run_batch(nets, lays, kw)
for b in range(2):
    ro = O.sqp_solve(O.problem_acopf(nets[b], lays[b]), O.default_options(kkt_mode=2, num_threads=16, **kw))
    tr = ctx.sqp_trace(b)
    print(b, "oracle", [a["ipm_iters"] for a in ro["trace"]])
    print(b, "device", [t["ipm_iters"] for t in tr])
    print(b, "fr    ", [t["fr"] for t in tr], "sub_status", [t["sub_status"] for t in tr])
