export TMPDIR=/tmp
mkdir -p gpurun_out
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_VMEM SQ_INSTS_LDS"; do
  tag=$(echo $set | cut -c1-12 | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pg_$tag -o p -- python tools/gemm_one.py ${N:-768} ${K:-3072} ${V:-81} > gpurun_out/pg_$tag.log 2>&1
  python tools/rocpd_pmc.py gpurun_out/pg_$tag/p_results.db 2>&1 | grep -E "kernel |gemm" | cut -c1-60,93-
  rm -rf gpurun_out/pg_$tag
done
