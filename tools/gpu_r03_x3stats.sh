# kernel breakdown of the strict (bf16x3) step at B = 256 after the attention core moved to split operands
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/px -o f -- python bench.py --dtype bf16x3 --batch 256 --steps 6 --warmup 2 --no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none > gpurun_out/px.log 2>&1
python tools/rocpd_summary.py gpurun_out/px/f_results.db > gpurun_out/r03_final_kernel_stats_bf16x3_b256.txt 2>&1; rm -rf gpurun_out/px
head -24 gpurun_out/r03_final_kernel_stats_bf16x3_b256.txt | cut -c1-200
tail -3 gpurun_out/px.log | cut -c1-300
