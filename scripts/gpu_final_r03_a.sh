#!/bin/bash
# Round-3 final artefacts, part A (one GPU): bench lines of the headline workload and its variants.  gpurun_out/final_r03/
R=$GRAFT_REPO_ROOT
F=$R/gpurun_out/final_r03
mkdir -p $F
cd $R
b() { out=$1; shift; timeout -k 10 600 python bench.py "$@" > $F/$out 2> $F/${out%.json}.err || { echo "FAILED $out"; tail -3 $F/${out%.json}.err; exit 1; }; python scripts/print_bench.py $F/$out; }
b r03_bench_steps20.json --steps 20 --warmup 5
b r03_bench_default.json
b r03_bench_textbook_sign.json --literal-quirks 0 --no-termination --no-dense-ldlt --no-screening --no-batch-curve
b r03_bench_acr_formulation.json --formulation acr --no-termination --no-dense-ldlt --no-screening --no-batch-curve
b r03_bench_batch64.json --batch 64 --steps 20 --warmup 5 --quick
SQPHIP_MF_TOP2=0 SQPHIP_MF_LEVEL2=0 SQPHIP_MF_BIG_LDSIMG=0 b r03_bench_old_solve_kernels.json --steps 20 --warmup 5 --quick
