export TMPDIR=/tmp
mkdir -p gpurun_out
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CU_CYCLES SQ_WAVES"; do
  tag=$(echo $set | cut -c1-14 | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pa_$tag -o p -- python tools/attn_bench.py ${ATTN_B:-512} > gpurun_out/pa_$tag.log 2>&1
  python tools/rocpd_pmc.py gpurun_out/pa_$tag/p_results.db 2>&1 | grep -E "kernel |attn" | cut -c1-50,93-
  rm -rf gpurun_out/pa_$tag
done
