#!/bin/bash
for b in 512 64; do
for c in 1 0; do
  v=$(timeout -k 10 120 python bench.py --steps 20 --warmup 5 --quick --batch $b --ipm-corrector $c 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); r=d['roofline']; print(round(d['value'],1), 'fac', d['config']['kkt_factorisations'], 'ipm', d['config']['ipm_iterations'], 'solves', r['instance_solves'], 'sweeps', d['config']['sweeps'], 'qp', d['config']['qp_solved'])")
  echo "batch $b corrector $c: $v"
done; done
