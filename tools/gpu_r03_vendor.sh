# the vendor library as a yardstick INSIDE the step: same bench command, nt_kernel 0 (ours) vs 200 (plain GEMMs through hipBLASLt)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels.py -q -k "variants_agree or specialised_epilogues" -p no:cacheprovider 2>&1 | tail -3
for k in 0 200 0 200; do
  timeout 200 python bench.py --nt-kernel $k --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --no-parity --strict-dtype none 2>gpurun_out/vendor_$k.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d['roofline']
print('nt_kernel=%d value %.1f ms %.3f gemm_ms %.2f loss %.4f' % ($k, d['value'], d['ms_per_step'], r['gemm_ms_per_step'], d['final_loss']))
for kk,v in r['by_kernel'].items(): print('   ', kk[:70], v)
"
done | tee gpurun_out/r03_vendor_in_step.txt
tail -3 gpurun_out/vendor_200.err
