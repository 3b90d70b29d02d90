# PMC passes over the grouped weight-gradient kernel (separate passes, kernel-trace only -- gpurun refuses more)
export TMPDIR=/tmp
mkdir -p gpurun_out
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INST_LEVEL_VMEM SQ_INSTS_LDS"; do
  tag=$(echo $set | cut -c1-12 | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pw_$tag -o p -- python tools/wgrad_bench.py 2 > gpurun_out/pw_$tag.log 2>&1
  python tools/rocpd_pmc.py gpurun_out/pw_$tag/p_results.db 2>&1 | grep -E "kernel |gemm_tn" | cut -c1-40,93-
  rm -rf gpurun_out/pw_$tag
done
