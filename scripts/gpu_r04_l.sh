#!/bin/bash
for c in 0 1; do
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --literal-quirks 0 --ipm-corrector $c --no-cpu-baseline --no-dense-ldlt --no-batch-curve --no-screening 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); t=d['termination']['literal_quirks_0']
print('lq0 corrector $c: timed', round(d['value'],1), 'QP/s fac/qp', round(d['config']['factorisations_per_qp'],2), '| termination leg:', round(t['qp_per_s'],1), 'QP/s', round(t['seconds'],2), 's converged', t['converged_ret0'], 'fac/qp', round(t['factorisations_per_qp'],2))"
done
