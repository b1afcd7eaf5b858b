#!/bin/bash
# Round-3 final artefacts, part C: one of the large shapes (convergent geographic networks): bench line, kernel stats.
# usage: gpu_final_r03_c.sh case1354|case9241
WL=$1
R=$GRAFT_REPO_ROOT
F=$R/gpurun_out/final_r03
mkdir -p $F
cd $R
timeout -k 10 800 python bench.py --workload $WL > $F/r03_bench_$WL.json 2> $F/r03_bench_$WL.err || { echo "FAILED"; tail -3 $F/r03_bench_$WL.err; exit 1; }
python scripts/print_bench.py $F/r03_bench_$WL.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $F/s$WL --output-format csv -- python3 $R/bench.py --workload $WL --quick > $F/r03_${WL}_under_rocprof.json 2> $F/s$WL.err || { tail -3 $F/s$WL.err; exit 1; }
cp $(find $F/s$WL -name '*kernel_stats.csv' | head -1) $F/r03_${WL}_kernel_stats.csv
rm -rf $F/s$WL
head -12 $F/r03_${WL}_kernel_stats.csv | cut -c1-150
