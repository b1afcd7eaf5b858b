#!/bin/bash
# Round-2 final artefacts (one GPU): bench lines of every workload / variant, rocprofv3 kernel stats + PMC traffic of the
# default command, the per-position timeline of a sweep.  Output: gpurun_out/final/ (copied into profiles/ by hand).
R=$GRAFT_REPO_ROOT
F=$R/gpurun_out/final
rm -rf $F && mkdir -p $F
cd $R
b() { out=$1; shift; timeout -k 10 600 python bench.py "$@" > $F/$out 2> $F/${out%.json}.err || { echo "FAILED $out"; tail -3 $F/${out%.json}.err; exit 1; }; python scripts/print_bench.py $F/$out; }
b r02_bench_default.json
b r02_bench_steps20.json --steps 20 --warmup 5
b r02_bench_textbook_sign.json --literal-quirks 0 --no-termination --no-dense-ldlt
b r02_bench_acr_formulation.json --formulation acr --no-termination --no-dense-ldlt
b r02_bench_dense_path_batch64.json --kkt-mode 1 --batch 64 --no-termination --no-dense-ldlt
b r02_bench_case1354.json --workload case1354
b r02_bench_case9241.json --workload case9241
bash scripts/gpu_profile_r02.sh r02 || exit 1
cp $R/gpurun_out/prof_r02/r02_bench_kernel_stats.csv $R/gpurun_out/prof_r02/r02_bench_under_rocprof.json $R/gpurun_out/prof_r02/r02_pmc_traffic.json $R/gpurun_out/prof_r02/mf_traffic.json $F/
rm -rf $R/gpurun_out/prof_r02
# timeline of one sweep: a single instance group, so that launch positions line up
cd /tmp && export TMPDIR=/tmp
SQPHIP_GROUPS=1 timeout -k 10 400 rocprofv3 --kernel-trace -d $F/trace --output-format csv -- python3 $R/scripts/gpu_sqp_run.py case118 512 6 2 > $F/trace_run.log 2> $F/trace.err || { tail -3 $F/trace.err; exit 1; }
python3 $R/scripts/trace_by_position.py $F/trace 60 160 > $F/r02_sweep_timeline.txt
rm -rf $F/trace
# kernel stats of the largest shape
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $F/s9241 --output-format csv -- python3 $R/bench.py --workload case9241 --no-cpu-baseline > $F/r02_case9241_under_rocprof.json 2> $F/s9241.err || { tail -3 $F/s9241.err; exit 1; }
cp $(find $F/s9241 -name '*kernel_stats.csv' | head -1) $F/r02_case9241_kernel_stats.csv
rm -rf $F/s9241
ls -la $F
