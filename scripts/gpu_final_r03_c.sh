#!/bin/bash
# Round-3 final artefacts, part C: the large shapes (convergent geographic networks): bench lines, kernel stats.
R=$GRAFT_REPO_ROOT
F=$R/gpurun_out/final_r03
mkdir -p $F
cd $R
b() { out=$1; shift; timeout -k 10 700 python bench.py "$@" > $F/$out 2> $F/${out%.json}.err || { echo "FAILED $out"; tail -3 $F/${out%.json}.err; exit 1; }; python scripts/print_bench.py $F/$out; }
b r03_bench_case1354.json --workload case1354
b r03_bench_case9241.json --workload case9241
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $F/s9241 --output-format csv -- python3 $R/bench.py --workload case9241 --quick > $F/r03_case9241_under_rocprof.json 2> $F/s9241.err || { tail -3 $F/s9241.err; exit 1; }
cp $(find $F/s9241 -name '*kernel_stats.csv' | head -1) $F/r03_case9241_kernel_stats.csv
rm -rf $F/s9241
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $F/s1354 --output-format csv -- python3 $R/bench.py --workload case1354 --quick > $F/r03_case1354_under_rocprof.json 2> $F/s1354.err || { tail -3 $F/s1354.err; exit 1; }
cp $(find $F/s1354 -name '*kernel_stats.csv' | head -1) $F/r03_case1354_kernel_stats.csv
rm -rf $F/s1354
ls -la $F
