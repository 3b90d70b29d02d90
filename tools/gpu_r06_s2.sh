# round 6, session 2: first split-K (plain slab stores + agent-scope release / acquire fences) + the decoder weight gradient on the small-token kernel: kernel tests, small-batch A/B, kernel traces at B = 8 / 16
export TMPDIR=/tmp
mkdir -p gpurun_out
START=$(date +%s)
timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/r06_s2_pytest_kernels.log 2>&1
echo "rc=$? wall=$(( $(date +%s) - START ))s" >> gpurun_out/r06_s2_pytest_kernels.log; tail -n 5 gpurun_out/r06_s2_pytest_kernels.log
bash tools/gpu_small_batch_ab.sh tools/libvisualbert_hip_ab_base.so 8 16 32 2>&1 | tee gpurun_out/r06_s2_small_batch_ab.txt
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
for B in 8 16; do
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --batch $B --steps 20 --warmup 5 $QUIET > gpurun_out/pf_b$B.log 2>&1
python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/r06_s2_kernel_stats_b$B.txt 2>&1; rm -rf gpurun_out/pf
head -n 12 gpurun_out/r06_s2_kernel_stats_b$B.txt | cut -c1-200
done
echo "total wall=$(( $(date +%s) - START ))s"
