# round 6, session 13: small-token weight gradients on 256x128 tiles with eight waves (in-tree) against the 128x128 kernel (no256)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider -k "wgrad" > gpurun_out/r06_s13_pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r06_s13_pytest.log; tail -n 4 gpurun_out/r06_s13_pytest.log
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
for B in 8 16 24 32; do for r in 1 2; do for lib in tools/libvisualbert_hip_ab_no256.so visualbert_amd/libvisualbert_hip.so; do
  timeout 300 python bench.py --batch $B --steps 30 --warmup 8 --lib-path $lib $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('B=%4d $lib: %.1f samples/s  %.3f ms/step (median %.3f)' % ($B, d['value'], d['ms_per_step'], d['ms_per_step_median']))" || tail -3 gpurun_out/ab.err
done; done; done 2>&1 | tee gpurun_out/r06_s13_wgrad256_ab.txt
for B in 8 32; do
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --batch $B --steps 20 --warmup 5 $QUIET > gpurun_out/pf_b$B.log 2>&1
python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/r06_s13_kernel_stats_b$B.txt 2>&1; rm -rf gpurun_out/pf
grep -E "^# kernels|gemm_tn" gpurun_out/r06_s13_kernel_stats_b$B.txt | cut -c1-200
done
