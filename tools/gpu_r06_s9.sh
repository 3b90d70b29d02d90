# round 6, session 9: kernel traces of the mid-size batches (B = 64, 128) on the current tree
export TMPDIR=/tmp
mkdir -p gpurun_out
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
for B in 64 128; do
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --batch $B --steps 20 --warmup 5 $QUIET > gpurun_out/pf_b$B.log 2>&1
python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/r06_s9_kernel_stats_b$B.txt 2>&1; rm -rf gpurun_out/pf
head -n 26 gpurun_out/r06_s9_kernel_stats_b$B.txt | cut -c1-200
done
