#!/bin/bash
# round 4: first runs of the spine kernel -- kernel tests, bit equality with the level launches, quick bench with and without it
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "multifrontal_kernels or every_front_kernel or spine_kernel or speculative_second or instance_groups or case118_scenarios_converge" > gpurun_out/r04_b_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_b_pytest.log
tail -4 gpurun_out/r04_b_pytest.log
grep -q "rc=0" gpurun_out/r04_b_pytest.log || exit 1
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --quick > gpurun_out/r04_b_bench_spine.json 2> gpurun_out/r04_b_bench_spine.err || exit 1
SQPHIP_MF_SPINE=0 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --quick > gpurun_out/r04_b_bench_nospine.json 2> gpurun_out/r04_b_bench_nospine.err || exit 1
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --quick --batch 64 > gpurun_out/r04_b_bench_spine_b64.json 2> gpurun_out/r04_b_bench_spine_b64.err || exit 1
SQPHIP_MF_SPINE=0 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --quick --batch 64 > gpurun_out/r04_b_bench_nospine_b64.json 2> gpurun_out/r04_b_bench_nospine_b64.err || exit 1
python - <<'PY'
import json
for f in ("spine","nospine","spine_b64","nospine_b64"):
    d=json.load(open(f"gpurun_out/r04_b_bench_{f}.json"))
    print(f, round(d["value"],1), round(d["ms_per_step"],2), d["config"]["kkt_factorisations"], round(d["roofline"]["frac"],4), round(d["roofline"]["factor_seconds"],3), round(d["roofline"]["solve_seconds"],3))
PY
