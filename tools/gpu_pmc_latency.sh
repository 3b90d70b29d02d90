# L1->L2 read latency and MFMA utilisation of the persistent NT GEMM at full chip vs a quarter of it (B=512 shapes):
# usage: bash tools/gpu_pmc_latency.sh   (inside a gpurun call; writes gpurun_out/pmc_latency.txt)
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/pmc_latency.txt
for shape in "768 3072" "2304 768"; do
  set -- $shape
  for wgs in 0 64; do
    for ctr in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
      tag=$(echo $ctr | cut -c1-10)
      M_ROWS=83968 WGS=$wgs rocprofv3 --kernel-trace --pmc $ctr -d gpurun_out/pl_$tag -o p -- python tools/gemm_one.py $1 $2 81 > gpurun_out/pl.log 2>&1
      echo "N=$1 K=$2 workgroups=${wgs/#0/all}" >> gpurun_out/pmc_latency.txt
      python tools/rocpd_pmc.py gpurun_out/pl_$tag/p_results.db 2>&1 | grep -E "kernel |gemm" | cut -c1-60,93- >> gpurun_out/pmc_latency.txt
      rm -rf gpurun_out/pl_$tag
    done
  done
done
cat gpurun_out/pmc_latency.txt
