# round 6, session 5: host-side call plans -- host profile at B = 8, then nosk / in-tree product builds at B = 8 / 16 / 32
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/host_profile.py --batch 8 > gpurun_out/r06_host_profile_b8_plans.txt 2>&1; head -30 gpurun_out/r06_host_profile_b8_plans.txt | cut -c1-150
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
for B in 8 16 32; do for r in 1 2; do for lib in tools/libvisualbert_hip_ab_nosk.so visualbert_amd/libvisualbert_hip.so; do
  timeout 300 python bench.py --batch $B --steps 30 --warmup 8 --lib-path $lib $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('B=%4d $lib: %.1f samples/s  %.3f ms/step (median %.3f)' % ($B, d['value'], d['ms_per_step'], d['ms_per_step_median']))" || tail -3 gpurun_out/ab.err
done; done; done 2>&1 | tee gpurun_out/r06_s5_small_batch_ab.txt
timeout 600 python -m pytest tests/test_model_parity.py tests/test_parity_at_scale.py -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/r06_s5_pytest_parity.log 2>&1
echo "rc=$?" >> gpurun_out/r06_s5_pytest_parity.log; tail -n 4 gpurun_out/r06_s5_pytest_parity.log
