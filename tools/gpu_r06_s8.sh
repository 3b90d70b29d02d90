# round 6, session 8: LDS pitches of the attention kernels' transposed tiles (8 -> 16 bytes of row padding: conflict-free 8-byte fragment reads)
#   attn_old = rounds 2-5 (TPAD 8, dS / K^T pitch 392) | attn_b = dS / K^T pitch 400 only | in-tree = every transposed bf16 tile padded by 16
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider -k "attention" > gpurun_out/r06_s8_pytest_attn.log 2>&1
echo "rc=$?" >> gpurun_out/r06_s8_pytest_attn.log; tail -n 4 gpurun_out/r06_s8_pytest_attn.log
for r in 1 2; do for lib in tools/libvisualbert_hip_ab_attn_old.so tools/libvisualbert_hip_ab_attn_b.so visualbert_amd/libvisualbert_hip.so; do
  echo "== $lib"; VB_LIB_PATH=$lib timeout 300 python tools/attn_bench.py 1024 164 2>&1 | grep -E "p=0.1" | grep -E "one-pass backward|fwd .* us \("
done; done 2>&1 | tee gpurun_out/r06_s8_attn_bench.txt
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
for r in 1 2 3; do for lib in tools/libvisualbert_hip_ab_attn_old.so tools/libvisualbert_hip_ab_attn_b.so visualbert_amd/libvisualbert_hip.so; do
  timeout 300 python bench.py --steps 15 --warmup 4 --lib-path $lib $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('$lib: %.1f samples/s  %.3f ms/step (median %.3f)' % (d['value'], d['ms_per_step'], d['ms_per_step_median']))" || tail -3 gpurun_out/ab.err
done; done 2>&1 | tee gpurun_out/r06_s8_attn_step_ab.txt
for lib in tools/libvisualbert_hip_ab_attn_old.so visualbert_amd/libvisualbert_hip.so; do
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --steps 8 --warmup 2 --lib-path $lib $QUIET > gpurun_out/pf.log 2>&1
  python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/r06_s8_stats_$(basename $lib .so).txt 2>&1; rm -rf gpurun_out/pf
  echo "$lib: $(head -1 gpurun_out/r06_s8_stats_$(basename $lib .so).txt)"; grep -E "attn_" gpurun_out/r06_s8_stats_$(basename $lib .so).txt | cut -c1-175 | head -5
done
