#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_parity_depth.py -x -q -m gpu -k "speculative or instance_groups or transition_period or scenario_queue or batched_sqp or flat_sparse or case118_scenarios or replays or bit_for_bit or every_front" > gpurun_out/r04_g_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_g_pytest.log
tail -5 gpurun_out/r04_g_pytest.log
grep -q "rc=0" gpurun_out/r04_g_pytest.log || exit 1
for rep in 1 2; do
for b in 512 64; do
  for e in X=0 SQPHIP_MF_INERTIA_KERNEL=1; do
  v=$(env $e timeout -k 10 120 python bench.py --steps 20 --warmup 5 --quick --batch $b 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],1), d['config']['sweeps'], round(d['config']['factorisations_per_qp'],2))")
  echo "batch $b $e: $v"
done; done; done
