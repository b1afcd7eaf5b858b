#!/bin/bash
# spine merge of the amalgamation (SQPHIP_SYM_CHAIN = largest merged front): QP/s of the default bench
mkdir -p gpurun_out
for cf in ${CHAINS:-0 96 128 160}; do
  echo "== chain $cf" >> gpurun_out/chain.log
  SQPHIP_SYM_CHAIN=$cf timeout -k 10 200 python bench.py --no-cpu-baseline --no-termination --no-dense-ldlt > gpurun_out/chain_$cf.json 2>/dev/null || { echo FAILED >> gpurun_out/chain.log; continue; }
  python scripts/print_bench.py gpurun_out/chain_$cf.json >> gpurun_out/chain.log
  SQPHIP_SYM_CHAIN=$cf timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-termination --no-dense-ldlt > gpurun_out/chain20_$cf.json 2>/dev/null && python scripts/print_bench.py gpurun_out/chain20_$cf.json >> gpurun_out/chain.log
done
cat gpurun_out/chain.log
