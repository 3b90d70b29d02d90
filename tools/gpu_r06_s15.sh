# round 6, session 15: the mid-size batches (B = 48 ... 96) with the NT GEMMs pinned to one pipelined kernel each (0 = the rule)
export TMPDIR=/tmp
mkdir -p gpurun_out
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
for B in 48 64 96; do for k in 0 22 42 24 90 81; do
  timeout 300 python bench.py --batch $B --steps 30 --warmup 8 --nt-kernel $k $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('B=%4d nt_kernel %3d: %.1f samples/s  %.3f ms/step (median %.3f)' % ($B, $k, d['value'], d['ms_per_step'], d['ms_per_step_median']))" || tail -3 gpurun_out/ab.err
done; done 2>&1 | tee gpurun_out/r06_s15_mid_batch_nt_kernel.txt
