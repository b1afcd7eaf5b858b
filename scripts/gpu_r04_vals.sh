#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "matrix_values_by or flat_sparse or speculative_second or scenario_queue" > gpurun_out/r04_e_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_e_pytest.log
tail -6 gpurun_out/r04_e_pytest.log
grep -q "rc=0" gpurun_out/r04_e_pytest.log || exit 1
for b in 512 256 128 64; do
for vi in 1 0; do
  v=$(SQPHIP_MF_VALS_INLINE=$vi timeout -k 10 120 python bench.py --steps 20 --warmup 5 --quick --batch $b 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],1), d['config']['kkt_factorisations'], d['config']['sweeps'])")
  echo "batch $b vals_inline $vi: $v"
done; done
for wl in case14 case1354; do
for vi in 1 0; do
  v=$(SQPHIP_MF_VALS_INLINE=$vi timeout -k 10 300 python bench.py --workload $wl --quick 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(round(d['value'],1))")
  echo "$wl vals_inline $vi: $v"
done; done
