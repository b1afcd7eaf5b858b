#!/bin/bash
# rocprofv3 kernel stats of the bench at a given resident batch (throughput-bound regime): usage gpu_stats_batch.sh BATCH
B=${1:-4096}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/stats_b$B
rm -rf $O && mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/s --output-format csv -- python3 $R/bench.py --batch $B --no-cpu-baseline --no-termination --no-dense-ldlt --no-screening > $O/bench.json 2> $O/err.log || { tail -3 $O/err.log; exit 1; }
cp $(find $O/s -name '*kernel_stats.csv' | head -1) $R/gpurun_out/kernel_stats_b$B.csv
rm -rf $O/s
python3 $R/scripts/print_bench.py $O/bench.json
