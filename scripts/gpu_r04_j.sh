#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dense_hessian or tile_order or dense_tile or qp_modes or condensed or published_qp" > gpurun_out/r04_j_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_j_pytest.log
tail -4 gpurun_out/r04_j_pytest.log
grep -q "rc=0" gpurun_out/r04_j_pytest.log || { tail -40 gpurun_out/r04_j_pytest.log; exit 1; }
bash scripts/gpu_r04_stats.sh dense --workload dense --quick 2>&1 | head -14
python - <<'PY'
import json
d=json.load(open("gpurun_out/stats_dense/bench_under_rocprof.json"))
print(round(d["value"],1), "QP/s", round(d["ms_per_step"],1), "ms/step", d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["share_of_wall"])
PY
