# the pre-training step at small per-GPU batches, two PRODUCT builds alternately: $1 = reference build, in-tree = candidate; $2.. = batches
export TMPDIR=/tmp
mkdir -p gpurun_out
OLD=${1:?usage: gpu_small_batch_ab.sh <reference libvisualbert_hip.so> [batch ...]}; shift
NEW=visualbert_amd/libvisualbert_hip.so
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
for B in ${@:-8 16 32 64 128}; do for r in 1 2; do for lib in $OLD $NEW; do
  timeout 300 python bench.py --batch $B --steps 30 --warmup 8 --lib-path $lib $QUIET > gpurun_out/ab.json 2>/dev/null
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('B=%4d $lib: %.1f samples/s  %.3f ms/step (median %.3f)' % ($B, d['value'], d['ms_per_step'], d['ms_per_step_median']))"
done; done; done
