# round 6, session 7: dropout + residual in the producing GEMM's epilogue -- parity at the bench shape, then product builds A/B at B = 1024
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_bench_shape.py tests/test_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider -k "layer or split_k or wgrad" > gpurun_out/r06_s7_pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r06_s7_pytest.log; tail -n 6 gpurun_out/r06_s7_pytest.log
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
for r in 1 2 3; do for lib in tools/libvisualbert_hip_ab_nofuse.so visualbert_amd/libvisualbert_hip.so tools/libvisualbert_hip_ab_fuse90.so; do
  timeout 300 python bench.py --steps 15 --warmup 4 --lib-path $lib $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('$lib: %.1f samples/s  %.3f ms/step (median %.3f)' % (d['value'], d['ms_per_step'], d['ms_per_step_median']))" || tail -3 gpurun_out/ab.err
done; done 2>&1 | tee gpurun_out/r06_s7_dropres_ab.txt
for lib in tools/libvisualbert_hip_ab_nofuse.so visualbert_amd/libvisualbert_hip.so; do
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --steps 8 --warmup 2 --lib-path $lib $QUIET > gpurun_out/pf.log 2>&1
  python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/r06_s7_stats_$(basename $lib .so).txt 2>&1; rm -rf gpurun_out/pf
  echo "$lib: $(head -1 gpurun_out/r06_s7_stats_$(basename $lib .so).txt)"; grep -E "gemm_nt|ln_fwd|ln_bwd" gpurun_out/r06_s7_stats_$(basename $lib .so).txt | cut -c1-175 | head -14
done
