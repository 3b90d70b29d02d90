# round 6, session 18: the overlapped BertAdam on a LOW-priority stream (VB_OV_PRIO=1; 0 = default priority as in session 17)
export TMPDIR=/tmp
mkdir -p gpurun_out
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
python -c "import torch; print('stream priority range', torch.cuda.Stream.priority_range())" 2>&1 | tail -1 | tee gpurun_out/r06_s18_optimizer_overlap_prio_ab.txt
timeout 900 python -m pytest tests/test_optimizer_overlap.py -m gpu -q --tb=short -p no:cacheprovider --timeout 600 2>&1 | tail -n 5 | tee gpurun_out/r06_s18_pytest_overlap.log
for r in 1 2; do for B in 8 16 32 64; do for f in "1" "0" "off"; do
  fl=""; [ $f = off ] && fl="--no-optimizer-overlap"
  VB_OV_PRIO=$f timeout 300 python bench.py --batch $B --steps 40 --warmup 8 $fl $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('B=%4d side-stream priority %-4s: %.1f samples/s  %.3f ms/step (median %.3f)' % ($B, '$f', d['value'], d['ms_per_step'], d['ms_per_step_median']))" || tail -3 gpurun_out/ab.err
done; done; done 2>&1 | tee -a gpurun_out/r06_s18_optimizer_overlap_prio_ab.txt
