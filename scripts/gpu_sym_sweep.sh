#!/bin/bash
# amalgamation parameters against the streamed solve kernels (quick bench lines, 512 x IEEE-118, driver's steps)
R=$GRAFT_REPO_ROOT
F=$R/gpurun_out/symsweep
rm -rf $F && mkdir -p $F
cd $R
run() { tag=$1; shift; env "$@" timeout -k 10 300 python bench.py --quick --steps 20 --warmup 5 > $F/$tag.json 2> $F/$tag.err && echo "$tag $(python scripts/print_bench.py $F/$tag.json)"; }
run base SQPHIP_X=0
run small24 SQPHIP_MF_SMALL_FRONT=24
run small48 SQPHIP_MF_SMALL_FRONT=48
run chain96 SQPHIP_SYM_CHAIN=96
run chain128 SQPHIP_SYM_CHAIN=128
run zf35 SQPHIP_MF_ZERO_FRAC=0.35
