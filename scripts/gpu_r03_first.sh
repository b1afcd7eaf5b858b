#!/bin/bash
# Round 3, first call of a session: GPU parity suite, the driver's bench command, kernel stats + sweep timeline of a
# quick run.  Output under gpurun_out/r03a/.
R=$GRAFT_REPO_ROOT
F=$R/gpurun_out/r03a
rm -rf $F && mkdir -p $F
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -s > $F/pytest_gpu.log 2>&1; rc=$?
grep -a "replay:\|passed\|failed" $F/pytest_gpu.log | tail -5
echo "pytest rc=$rc"; [ $rc -ge 124 ] && exit $rc
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $F/bench_steps20.json 2> $F/bench_steps20.err || { tail -5 $F/bench_steps20.err; exit 1; }
python scripts/print_bench.py $F/bench_steps20.json
bash scripts/gpu_stats_quick.sh r03a --steps 20 --warmup 5
