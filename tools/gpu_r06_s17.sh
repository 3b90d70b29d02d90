# round 6, session 17: the overlapped BertAdam with the cheaper hook (raw stream handle, one event per layer) against the one-pass
# step at the small batches, alternating on one box
export TMPDIR=/tmp
mkdir -p gpurun_out
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
timeout 900 python -m pytest tests/test_optimizer_overlap.py -m gpu -q --tb=short -p no:cacheprovider --timeout 600 2>&1 | tail -n 15 | tee gpurun_out/r06_s17_pytest_overlap.log
for r in 1 2 3; do for B in 8 12 16 24 32; do for f in "" "--no-optimizer-overlap"; do
  timeout 300 python bench.py --batch $B --steps 40 --warmup 8 $f $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('B=%4d %-24s: %.1f samples/s  %.3f ms/step (median %.3f)' % ($B, '$f' or 'overlapped', d['value'], d['ms_per_step'], d['ms_per_step_median']))" || tail -3 gpurun_out/ab.err
done; done; done 2>&1 | tee gpurun_out/r06_s17_optimizer_overlap_ab.txt
python tools/host_profile.py --batch 8 > gpurun_out/r06_s17_host_profile_b8.txt 2>&1; tail -n 25 gpurun_out/r06_s17_host_profile_b8.txt
