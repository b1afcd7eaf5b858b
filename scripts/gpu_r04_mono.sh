#!/bin/bash
mkdir -p gpurun_out
for b in 512 256 128 64; do
  v=$(timeout -k 10 120 python bench.py --steps 20 --warmup 5 --quick --batch $b 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print(round(d['value'],1), 'fac/qp', round(d['config']['factorisations_per_qp'],2), 'sweeps', d['config']['sweeps'], 'frac', round(r['frac'],4))")
  echo "batch $b: $v"
done
timeout -k 10 120 python bench.py --quick | python -c "import json,sys; d=json.load(sys.stdin); print('defaults:', round(d['value'],1))"
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/r04_f_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_f_pytest.log
tail -30 gpurun_out/r04_f_pytest.log
