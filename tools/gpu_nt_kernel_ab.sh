# in-step A/B of the GEMM dispatch: default rule against everything pinned to the persistent kernel (81) / the two-workgroup kernel (90)
export TMPDIR=/tmp
mkdir -p gpurun_out
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off"
for r in 1 2 3; do for k in 0 81 90; do
  timeout 300 python bench.py --steps 15 --warmup 4 --nt-kernel $k $QUIET > gpurun_out/ab.json 2>/dev/null
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('nt_kernel $k: %.1f samples/s  %.3f ms/step (median %.3f)' % (d['value'], d['ms_per_step'], d['ms_per_step_median']))"
done; done
