#!/bin/bash
# round 4: flat sparse products -- bit equality test, then the large shapes with and without them
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "flat_sparse or spine_kernel or transition_period or qp_modes_case14 or subproblems_on_case1354" > gpurun_out/r04_d_pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r04_d_pytest.log
tail -6 gpurun_out/r04_d_pytest.log
grep -q "rc=0" gpurun_out/r04_d_pytest.log || exit 1
for wl in case1354 case9241; do
for fl in 1 0; do
  SQPHIP_VEC_FLAT=$fl timeout -k 10 400 python bench.py --workload $wl --quick > gpurun_out/r04_d_${wl}_flat$fl.json 2> gpurun_out/r04_d_${wl}_flat$fl.err || { tail -3 gpurun_out/r04_d_${wl}_flat$fl.err; exit 1; }
  python -c "
import json
d=json.load(open('gpurun_out/r04_d_${wl}_flat$fl.json'))
print('$wl flat $fl:', round(d['value'],2), 'QP/s', round(d['ms_per_step'],1), 'ms/step', d['config']['kkt_factorisations'], round(d['roofline']['frac'],4))"
done; done
python bench.py --steps 20 --warmup 5 --quick | python -c "import json,sys; d=json.load(sys.stdin); print('case118:', round(d['value'],1))"
