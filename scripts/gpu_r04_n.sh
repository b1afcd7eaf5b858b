#!/bin/bash
# A/B on one box: the round-3 tree (alt/r03, built beforehand) against this tree, termination legs and the driver's command
R=$GRAFT_REPO_ROOT
show() { python -c "
import json,sys
d=json.load(sys.stdin); t=d['termination']
for k,v in t.items(): print('$1', k, round(v['qp_per_s'],1), 'QP/s', round(v['seconds'],2), 's qp', v['qp_solved'], 'conv', v['converged_ret0'])
print('$1 timed:', round(d['value'],1))"; }
(cd $R/alt/r03 && timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dense-ldlt --no-batch-curve --no-screening 2>/dev/null | show r03)
(cd $R && timeout -k 10 300 python bench.py --steps 20 --warmup 5 --ipm-corrector 1 --no-cpu-baseline --no-dense-ldlt --no-batch-curve --no-screening 2>/dev/null | show r04-mpc)
(cd $R && timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dense-ldlt --no-batch-curve --no-screening 2>/dev/null | show r04-mono)
