# round 6, session 16: BertAdam stepping the encoder layers behind the backward pass (second stream) against the one-pass step,
# alternating on one box, B = 8 ... 1024; then the new tests and the data-parallel tests that now run with it
export TMPDIR=/tmp
mkdir -p gpurun_out
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
timeout 900 python -m pytest tests/test_optimizer_overlap.py tests/test_model_parity.py -m gpu -q --tb=short -p no:cacheprovider --timeout 600 2>&1 | tail -n 6 | tee gpurun_out/r06_s16_pytest_overlap.log
for r in 1 2; do for B in 8 16 32 64 128 1024; do for f in "" "--no-optimizer-overlap"; do
  st=40; [ $B -ge 128 ] && st=20; [ $B -ge 1024 ] && st=12
  timeout 300 python bench.py --batch $B --steps $st --warmup 8 $f $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('B=%4d %-24s: %.1f samples/s  %.3f ms/step (median %.3f)' % ($B, '$f' or 'overlapped', d['value'], d['ms_per_step'], d['ms_per_step_median']))" || tail -3 gpurun_out/ab.err
done; done; done 2>&1 | tee gpurun_out/r06_s16_optimizer_overlap_ab.txt
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --batch 8 --steps 20 --warmup 6 $QUIET > gpurun_out/pf.log 2>&1
python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/r06_s16_kernel_stats_b8.txt 2>&1; rm -rf gpurun_out/pf
head -n 12 gpurun_out/r06_s16_kernel_stats_b8.txt | cut -c1-200
timeout 1200 python -m pytest tests/test_data_parallel_cpu.py tests/test_bench_launch.py -m gpu -q --tb=short -p no:cacheprovider --timeout 900 2>&1 | tail -n 6 | tee gpurun_out/r06_s16_pytest_dp.log
