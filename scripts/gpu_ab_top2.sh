#!/bin/bash
# A/B of the streamed top-of-tree solve (k_mf_solve_top2) against k_mf_solve_top: kernel tests, then quick bench lines.
R=$GRAFT_REPO_ROOT
F=$R/gpurun_out/top2
rm -rf $F && mkdir -p $F
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "multifrontal or subproblems_on_case1354 or case118_scenarios_converge or qp_modes" > $F/pytest.log 2>&1; rc=$?
tail -3 $F/pytest.log
[ $rc -eq 0 ] || exit $rc
for v in 1 0; do
  SQPHIP_MF_TOP2=$v timeout -k 10 300 python bench.py --quick --steps 20 --warmup 5 > $F/bench_top2_$v.json 2> $F/bench_top2_$v.err || { tail -5 $F/bench_top2_$v.err; exit 1; }
  python scripts/print_bench.py $F/bench_top2_$v.json
done
SQPHIP_MF_TOP2=1 timeout -k 10 300 python bench.py --quick --steps 20 --warmup 5 --batch 64 > $F/bench_top2_b64.json 2> $F/b64.err && python scripts/print_bench.py $F/bench_top2_b64.json
