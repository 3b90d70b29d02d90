# round 6, session 21: the layer's three second-stage reductions (LayerNorm x 2, attention bias gradient) as ONE launch at the end of the
# layer's backward, against one launch each (tools/build_variant.sh nodefer "-DVB_DEFER_REDUCE=0"), alternating on one box
export TMPDIR=/tmp
mkdir -p gpurun_out
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
for r in 1 2 3; do for B in 8 16 32 128; do for arm in defer nodefer; do
  lp=""; [ $arm = nodefer ] && lp="--lib-path tools/libvisualbert_hip_ab_nodefer.so"
  st=40; [ $B -ge 128 ] && st=20
  timeout 300 python bench.py --batch $B --steps $st --warmup 8 $lp $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('B=%4d %-8s: %.1f samples/s  %.3f ms/step (median %.3f)' % ($B, '$arm', d['value'], d['ms_per_step'], d['ms_per_step_median']))" || tail -3 gpurun_out/ab.err
done; done; done 2>&1 | tee gpurun_out/r06_s21_defer_reduce_ab.txt
timeout 300 python bench.py --steps 15 --warmup 4 $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err; python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('B=1024 defer: %.1f samples/s %.3f ms' % (d['value'], d['ms_per_step']))" | tee -a gpurun_out/r06_s21_defer_reduce_ab.txt
timeout 300 python bench.py --steps 15 --warmup 4 --lib-path tools/libvisualbert_hip_ab_nodefer.so $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err; python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('B=1024 nodefer: %.1f samples/s %.3f ms' % (d['value'], d['ms_per_step']))" | tee -a gpurun_out/r06_s21_defer_reduce_ab.txt
timeout 1500 python -m pytest tests/test_kernels.py tests/test_model_parity.py tests/test_parity_at_scale.py tests/test_bench_shape.py -m gpu -q --tb=short -p no:cacheprovider --timeout 900 > gpurun_out/r06_s21_pytest.log 2>&1; tail -n 6 gpurun_out/r06_s21_pytest.log
