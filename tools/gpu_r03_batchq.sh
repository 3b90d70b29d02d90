# wave-quantisation probe: per-GPU batch sizes whose GEMM tile counts fill whole rounds of the 512 resident workgroups
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels.py tests/test_bf16x3.py -q -k "wgrad" -p no:cacheprovider 2>&1 | tail -2
for b in ${BATCHES:-1024 1061 1024 1061 998 1000}; do
  timeout 200 python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --no-parity --strict-dtype none 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('B=%d value %.1f ms %.3f dom_us %.1f gemm_ms %.2f' % ($b, d['value'], d['ms_per_step'], r['avg_launch_us'], r['gemm_ms_per_step']))
for k,v in r['by_kernel'].items(): print('   ', k[:60], v)
"
done | tee gpurun_out/r03_batch_quantisation.txt
