#!/bin/bash
# Round-4 final artefacts, part A (one GPU): the GPU test suite, then bench lines of the headline workload and its variants.
R=$GRAFT_REPO_ROOT
F=$R/gpurun_out/final_r04
mkdir -p $F
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -q > $F/r04_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> $F/r04_pytest_gpu.log
tail -3 $F/r04_pytest_gpu.log
b() { out=$1; shift; timeout -k 10 600 python bench.py "$@" > $F/$out 2> $F/${out%.json}.err || { echo "FAILED $out"; tail -3 $F/${out%.json}.err; exit 1; }; python scripts/print_bench.py $F/$out; }
b r04_bench_steps20.json --steps 20 --warmup 5
b r04_bench_default.json
b r04_bench_mehrotra.json --steps 20 --warmup 5 --ipm-corrector 1 --no-termination --no-dense-ldlt --no-screening --no-batch-curve --no-cpu-baseline
b r04_bench_textbook_sign.json --literal-quirks 0 --no-termination --no-dense-ldlt --no-screening --no-batch-curve
b r04_bench_acr_formulation.json --formulation acr --no-termination --no-dense-ldlt --no-screening --no-batch-curve
b r04_bench_batch64.json --batch 64 --steps 20 --warmup 5 --quick
b r04_bench_batch128.json --batch 128 --steps 20 --warmup 5 --quick
