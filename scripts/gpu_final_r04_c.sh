#!/bin/bash
# Round-4 final artefacts, part C: one of the large shapes: bench line, kernel stats, PMC traffic.  usage: gpu_final_r04_c.sh case1354|case9241
WL=$1
R=$GRAFT_REPO_ROOT
F=$R/gpurun_out/final_r04
mkdir -p $F
cd $R
timeout -k 10 900 python bench.py --workload $WL > $F/r04_bench_$WL.json 2> $F/r04_bench_$WL.err || { echo "FAILED"; tail -3 $F/r04_bench_$WL.err; exit 1; }
python scripts/print_bench.py $F/r04_bench_$WL.json
bash scripts/gpu_profile.sh r04 $WL --quick || exit 1
cp $R/gpurun_out/prof_r04_$WL/r04_* $R/gpurun_out/prof_r04_$WL/mf_traffic_$WL.json $F/
head -12 $F/r04_bench_${WL}_kernel_stats.csv | cut -c1-150
