#!/bin/bash
# only the two --pmc passes of gpu_profile_r02.sh (FETCH_SIZE, WRITE_SIZE) + the post-processing; the kernel-stats pass
# and its bench line are taken from gpurun_out/final/ (scripts/gpu_profile_final.sh)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_pmc
rm -rf $O && mkdir -p $O/stats
cp $R/profiles/r02_bench_under_rocprof.json $O/bench_under_rocprof.json
PMCARGS="--steps 2 --warmup 0 --no-termination --no-cpu-baseline --no-dense-ldlt --no-screening --no-kernel-timing"
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py $PMCARGS > $O/fetch.json 2> $O/fetch.err || { tail -5 $O/fetch.err; exit 1; }
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/bench.py $PMCARGS > $O/write.json 2> $O/write.err || { tail -5 $O/write.err; exit 1; }
python3 $R/scripts/make_profile_r02.py $O r02
mkdir -p $R/gpurun_out/pmc && cp $O/r02_pmc_traffic.json $O/mf_traffic.json $R/gpurun_out/pmc/
rm -rf $O
