# L2-miss traffic (FETCH_SIZE / WRITE_SIZE, separate passes) per GEMM shape at M = 167,936, next to the algorithmic bytes:
# which shapes re-fetch their operands.   -> gpurun_out/r03_pmc_fetch_by_shape.txt
export TMPDIR=/tmp
mkdir -p gpurun_out
export M_ROWS=167936
out=gpurun_out/r03_pmc_fetch_by_shape.txt
: > $out
run() {  # tag, N, K, variant
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pf_$1_$c -o p -- python tools/gemm_one.py $2 $3 $4 > gpurun_out/pf_$1_$c.log 2>&1
    echo "## $1 N=$2 K=$3 variant=$4 EPI=${EPI:-plain} $c" >> $out
    python tools/rocpd_pmc.py gpurun_out/pf_$1_$c/p_results.db 2>&1 | grep -E "kernel |gemm_nt" | cut -c1-60,93- >> $out
    rm -rf gpurun_out/pf_$1_$c
  done
}
run qkv 2304 768 90
run attnout 768 768 90
EPI=gelu run ffnin 3072 768 90
EPI=add run ffnin_dgrad 768 3072 90
run ffnout 768 3072 81
cat $out
