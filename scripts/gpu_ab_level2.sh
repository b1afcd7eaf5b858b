#!/bin/bash
# A/B of the LDS-staged level-launch solves (k_mf_fwd2 / k_mf_bwd2) against k_mf_fwd / k_mf_bwd: kernel tests, quick bench lines.
R=$GRAFT_REPO_ROOT
F=$R/gpurun_out/lvl2
rm -rf $F && mkdir -p $F
cd $R
rc=0

[ $rc -eq 0 ] || exit $rc
for v in 1 0; do
  SQPHIP_MF_LEVEL2=$v timeout -k 10 300 python bench.py --quick --steps 20 --warmup 5 > $F/bench_$v.json 2> $F/bench_$v.err || { tail -5 $F/bench_$v.err; exit 1; }
  python scripts/print_bench.py $F/bench_$v.json
done
for v in 1 0; do
SQPHIP_MF_LEVEL2=$v timeout -k 10 300 python bench.py --quick --steps 20 --warmup 5 --batch 64 > $F/bench_b64_$v.json 2> $F/b64.err && python scripts/print_bench.py $F/bench_b64_$v.json
done
for v in 1 0; do
SQPHIP_MF_LEVEL2=$v timeout -k 10 300 python bench.py --quick --workload case1354 > $F/bench_1354_$v.json 2> $F/b1354.err && python scripts/print_bench.py $F/bench_1354_$v.json
done
