# shader clock / power while (a) the bulk Schur-update kernel and (b) the pure fp64 MFMA probe hold the chip
./scripts/probes/trailing_trace 4 1 ${REPS:-4000} ${RANDOM_DATA:-1} > gpurun_out/clk2_trailing.log 2>&1 &
BP=$!
sleep 3
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 1; done > gpurun_out/clk2_smi_trailing.log 2>&1
wait $BP
rocm-smi --showmaxpower 2>&1 | grep -i "max" > gpurun_out/clk2_maxpower.log
cat gpurun_out/clk2_trailing.log | head -3; cat gpurun_out/clk2_smi_trailing.log gpurun_out/clk2_maxpower.log
