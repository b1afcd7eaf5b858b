# PMC passes over the LDL^T micro-benchmark (one counter group per run; no tracing flags)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" \
           "TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum" \
           "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp -d $R/gpurun_out/pmc_ldlt/p$i --output-format csv -- python3 $R/scripts/gpu_ldlt_bench.py 2813 64 2 > $R/gpurun_out/pmc_ldlt_p$i.log 2>&1 || { tail -5 $R/gpurun_out/pmc_ldlt_p$i.log; exit 1; }
done
python3 $R/scripts/pmc_summary.py $R/gpurun_out/pmc_ldlt k_trailing
