export TMPDIR=/tmp
mkdir -p gpurun_out
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off"
for B in 8 64; do
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --batch $B --steps 20 --warmup 5 $QUIET > gpurun_out/pf_b$B.log 2>&1
python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/r05_kernel_stats_b$B.txt 2>&1; rm -rf gpurun_out/pf
head -n 3 gpurun_out/r05_kernel_stats_b$B.txt | cut -c1-160
tail -n 1 gpurun_out/pf_b$B.log | cut -c1-200
done
