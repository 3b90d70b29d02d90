# round 6, session 14: crossover of the 256x128 small-token kernel against the persistent kernel: 96 (in-tree) / 128 / 192 K tiles at B = 40 ... 64
export TMPDIR=/tmp
mkdir -p gpurun_out
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
for B in 8 32 40 48 64; do for r in 1 2; do for lib in visualbert_amd/libvisualbert_hip.so tools/libvisualbert_hip_ab_kt128.so tools/libvisualbert_hip_ab_kt192.so; do
  timeout 300 python bench.py --batch $B --steps 30 --warmup 8 --lib-path $lib $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('B=%4d $lib: %.1f samples/s  %.3f ms/step (median %.3f)' % ($B, d['value'], d['ms_per_step'], d['ms_per_step_median']))" || tail -3 gpurun_out/ab.err
done; done; done 2>&1 | tee gpurun_out/r06_s14_wgrad256_crossover.txt
