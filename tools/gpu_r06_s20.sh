# round 6, session 20: polynomial GELU pair, per GEMM and per kernel (81 = persistent 256x256, 90 = two-workgroup 256x128): in-tree library
# (VB_GELU_POLY=1) against tools/libvisualbert_hip_ab_nopoly.so, alternating processes; then the step A/B
export TMPDIR=/tmp
mkdir -p gpurun_out
QUIET="--no-cpu-baseline --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg --no-profile"
for r in 1 2; do for arm in poly nopoly; do
  echo "== $arm (round $r)"
  if [ $arm = nopoly ]; then export VB_LIB_PATH=tools/libvisualbert_hip_ab_nopoly.so; else unset VB_LIB_PATH; fi
  VB_NOCHECK=1 timeout 300 python tools/gemm_ab.py 1024 81 90 2>&1 | grep -i "gelu\|per step"
done; done 2>&1 | tee gpurun_out/r06_s20_gelu_poly_gemm_ab.txt
unset VB_LIB_PATH
for r in 1 2 3; do for arm in poly nopoly; do
  lp=""; [ $arm = nopoly ] && lp="--lib-path tools/libvisualbert_hip_ab_nopoly.so"
  timeout 300 python bench.py --steps 15 --warmup 4 $lp $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "
import json;d=json.load(open('gpurun_out/ab.json'))
print('%-7s: %.1f samples/s  %.3f ms/step (median %.3f)' % ('$arm', d['value'], d['ms_per_step'], d['ms_per_step_median']))" || tail -3 gpurun_out/ab.err
done; done 2>&1 | tee gpurun_out/r06_s20_gelu_poly_step_ab.txt
