#!/bin/bash
for n in 1920 2688; do
timeout -k 10 600 python bench.py --workload dense --dense-n $n --quick 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); r=d['roofline']
print('dense n=$n:', round(d['value'],1), 'QP/s', round(d['ms_per_step'],1), 'ms/step  update kernels', round(r['achieved'],2), 'TFLOP/s', round(r['frac'],3), 'share', round(r['share_of_wall'],3))"
done
