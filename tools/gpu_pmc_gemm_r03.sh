# PMC passes (separate runs, kernel-trace only) over the GEMM kernels that SHIP, at the bench's M = 167,936 tokens:
#   k90 plain (QKV projection), k90 with the GELU + GELU' epilogue (FFN-in forward), k81 (FFN-out forward), grouped wgrad (one encoder layer).  usage: bash tools/gpu_pmc_gemm_r03.sh  -> gpurun_out/r03_pmc_gemm_*.txt
export TMPDIR=/tmp
mkdir -p gpurun_out
SETS=("SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS")
run() {  # tag, grep pattern, command...
  tag=$1; pat=$2; shift 2
  out=gpurun_out/r03_pmc_gemm_$tag.txt
  echo "# $*" > $out
  i=0
  for set in "${SETS[@]}"; do
    rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pq_$tag$i -o p -- "$@" > gpurun_out/pq_$tag$i.log 2>&1
    python tools/rocpd_pmc.py gpurun_out/pq_$tag$i/p_results.db 2>&1 | grep -E "kernel |$pat" | cut -c1-70,93- >> $out
    rm -rf gpurun_out/pq_$tag$i
    i=$((i+1))
  done
  cat $out
}
export M_ROWS=167936
run k90_qkv gemm_nt_dual python tools/gemm_one.py 2304 768 90
EPI=gelu run k90_ffnin_gelu gemm_nt_dual python tools/gemm_one.py 3072 768 90
run k81_ffnout gemm_nt_8ph python tools/gemm_one.py 768 3072 81
VB_TOKENS=167936 run tn_wgrad gemm_tn python tools/wgrad_bench.py 2
