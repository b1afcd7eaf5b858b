#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_depth.py -x -q -m gpu -s -k "dense_hessian or case9241_scenario or acceptable" > gpurun_out/r04_i_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_i_pytest.log
grep -E "9241 fixture|sub-problem|passed|failed|rc=" gpurun_out/r04_i_pytest.log | tail -14
grep -q "rc=0" gpurun_out/r04_i_pytest.log || { tail -40 gpurun_out/r04_i_pytest.log; exit 1; }
timeout -k 10 600 python bench.py --workload dense --no-cpu-baseline > gpurun_out/r04_i_dense.json 2> gpurun_out/r04_i_dense.err || { tail -5 gpurun_out/r04_i_dense.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_i_dense.json"))
print(round(d["value"],1), "QP/s", round(d["ms_per_step"],1), "ms/step", d["config"]["qp_solved"], d["config"]["ipm_iterations_per_qp"], d["config"]["instances_done_in_timed_steps"])
print(d["roofline"])
PY
