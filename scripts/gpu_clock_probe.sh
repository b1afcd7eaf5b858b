# samples sclk / power with rocm-smi while the LDL^T micro-benchmark runs (is the fp64 MFMA clock throttled?)
SQPHIP_OUTER=${SQPHIP_OUTER:-4} timeout -k 10 150 python scripts/gpu_ldlt_bench.py 2813 64 2500 > gpurun_out/clk_bench.log 2>&1 &
BP=$!
sleep 8
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 2; done > gpurun_out/clk_smi.log 2>&1
wait $BP
cat gpurun_out/clk_bench.log
cat gpurun_out/clk_smi.log
