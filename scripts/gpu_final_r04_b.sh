#!/bin/bash
# Round-4 final artefacts, part B: rocprofv3 kernel stats + PMC traffic of the driver's command (case118), with the level
# launches (default) and with the spine kernel (SQPHIP_MF_SPINE=1); the sweep timeline of one instance group; batch 64 stats;
# the small / dense workloads' bench lines.
R=$GRAFT_REPO_ROOT
F=$R/gpurun_out/final_r04
mkdir -p $F
cd $R
bash scripts/gpu_profile.sh r04 case118 --steps 20 --warmup 5 --no-cpu-baseline --no-termination --no-dense-ldlt --no-screening --no-batch-curve || exit 1
cp $R/gpurun_out/prof_r04_case118/r04_* $R/gpurun_out/prof_r04_case118/mf_traffic_case118.json $F/
SQPHIP_MF_SPINE=1 bash scripts/gpu_profile.sh r04spine case118 --steps 20 --warmup 5 --quick || exit 1
cp $R/gpurun_out/prof_r04spine_case118/mf_traffic_case118.json $F/mf_traffic_case118_spine_kernel.json
cp $R/gpurun_out/prof_r04spine_case118/r04spine_bench_case118_kernel_stats.csv $F/r04_bench_case118_spine_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
SQPHIP_GROUPS=1 timeout -k 10 400 rocprofv3 --kernel-trace -d $F/trace --output-format csv -- python3 $R/scripts/gpu_sqp_run.py case118 512 6 2 > $F/trace_run.log 2> $F/trace.err || { tail -3 $F/trace.err; exit 1; }
python3 $R/scripts/trace_by_position.py $F/trace 60 160 > $F/r04_sweep_timeline.txt
rm -rf $F/trace
cd $R
bash scripts/gpu_r04_stats.sh c118_b64_final --steps 20 --warmup 5 --quick --batch 64 > /dev/null && cp gpurun_out/stats_c118_b64_final/kernel_stats.csv $F/r04_bench_batch64_kernel_stats.csv
b() { out=$1; shift; timeout -k 10 600 python bench.py "$@" > $F/$out 2> $F/${out%.json}.err || { echo "FAILED $out"; tail -3 $F/${out%.json}.err; exit 1; }; python scripts/print_bench.py $F/$out; }
b r04_bench_case14.json --workload case14
b r04_bench_dense.json --workload dense
b r04_bench_dense_n1920.json --workload dense --dense-n 1920 --no-cpu-baseline
b r04_bench_dense_path_batch64.json --kkt-mode 1 --batch 64 --no-termination --no-dense-ldlt --no-screening --no-batch-curve
bash scripts/gpu_r04_stats.sh dense_final --workload dense --quick > /dev/null && cp gpurun_out/stats_dense_final/kernel_stats.csv $F/r04_dense_kernel_stats.csv
