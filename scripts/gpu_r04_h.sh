#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_depth.py -x -q -m gpu -s -k "no_hessian or multifrontal_kernels or replays or dropin_seat or elastic_step" > gpurun_out/r04_h_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_h_pytest.log
grep -E "replay:|passed|failed|rc=" gpurun_out/r04_h_pytest.log | tail -6
grep -q "rc=0" gpurun_out/r04_h_pytest.log || { tail -40 gpurun_out/r04_h_pytest.log; exit 1; }
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-termination --no-screening --no-cpu-baseline --no-dense-ldlt --no-batch-curve > gpurun_out/r04_h_bench.json 2> gpurun_out/r04_h_bench.err || { tail -5 gpurun_out/r04_h_bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_h_bench.json"))
print(round(d["value"],1), d["roofline"]["frac"])
for k,v in (d["kernels"] or {}).items():
    print(k, {a: (round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ("units","avg_us_per_launch_group","achieved","frac","share_of_timed_kernel_seconds","launch_groups")})
PY
