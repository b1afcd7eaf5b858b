#!/bin/bash
# A/B of two builds of the library on quick bench lines: the in-tree one against ALT (a .so inside the snapshot).
# usage: gpu_ab_lib.sh ALT.so [bench args]
ALT=$1; shift
R=$GRAFT_REPO_ROOT
F=$R/gpurun_out/ab_lib
rm -rf $F && mkdir -p $F
cd $R
cat > $F/run_alt.py <<PY
import sys, runpy
sys.path.insert(0, "$R")
from sqpsolver_jl_amd import _lib
_lib.SO_PATH = "$R/$ALT"
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path("$R/bench.py", run_name="__main__")
PY
for v in tree alt tree alt; do
  if [ $v = tree ]; then cmd="python bench.py"; else cmd="python $F/run_alt.py"; fi
  timeout -k 10 300 $cmd --quick --steps 20 --warmup 5 "$@" > $F/bench_$v.json 2> $F/bench_$v.err || { tail -5 $F/bench_$v.err; exit 1; }
  echo "$v $(python scripts/print_bench.py $F/bench_$v.json)"
done
