#!/bin/bash
# A/B of one environment switch on quick bench lines.  usage: gpu_ab_env.sh VAR [bench args]
V=$1; shift
R=$GRAFT_REPO_ROOT
F=$R/gpurun_out/ab_$V
rm -rf $F && mkdir -p $F
cd $R
for v in 1 0 1 0; do
  env $V=$v timeout -k 10 300 python bench.py --quick --steps 20 --warmup 5 "$@" > $F/bench_$v.json 2> $F/bench_$v.err || { tail -5 $F/bench_$v.err; exit 1; }
  echo "$V=$v $(python scripts/print_bench.py $F/bench_$v.json)"
done
