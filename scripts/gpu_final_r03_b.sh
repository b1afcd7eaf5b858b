#!/bin/bash
# Round-3 final artefacts, part B: rocprofv3 kernel stats + PMC traffic of the driver's command (case118), the sweep
# timeline of one instance group, the other workloads' bench lines.
R=$GRAFT_REPO_ROOT
F=$R/gpurun_out/final_r03
mkdir -p $F
cd $R
bash scripts/gpu_profile.sh r03 case118 --steps 20 --warmup 5 --no-cpu-baseline --no-termination --no-dense-ldlt --no-screening --no-batch-curve || exit 1
cp $R/gpurun_out/prof_r03_case118/r03_* $R/gpurun_out/prof_r03_case118/mf_traffic_case118.json $F/
cd /tmp && export TMPDIR=/tmp
SQPHIP_GROUPS=1 timeout -k 10 400 rocprofv3 --kernel-trace -d $F/trace --output-format csv -- python3 $R/scripts/gpu_sqp_run.py case118 512 6 2 > $F/trace_run.log 2> $F/trace.err || { tail -3 $F/trace.err; exit 1; }
python3 $R/scripts/trace_by_position.py $F/trace 60 160 > $F/r03_sweep_timeline.txt
rm -rf $F/trace
cd $R
b() { out=$1; shift; timeout -k 10 600 python bench.py "$@" > $F/$out 2> $F/${out%.json}.err || { echo "FAILED $out"; tail -3 $F/${out%.json}.err; exit 1; }; python scripts/print_bench.py $F/$out; }
b r03_bench_case14.json --workload case14
b r03_bench_dense_path_batch64.json --kkt-mode 1 --batch 64 --no-termination --no-dense-ldlt --no-screening --no-batch-curve
