# in-step A/B of the GEMM dispatch: default rule (0) against every K-contiguous GEMM pinned to the persistent kernel (81) / the two-workgroup
# kernel (90); $1 = dtype (bf16 | bf16x3), default bf16
export TMPDIR=/tmp
mkdir -p gpurun_out
DT=${1:-bf16}
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off"
for r in 1 2 3; do for k in 0 81 90; do
  timeout 300 python bench.py --dtype $DT --batch 1024 --steps 12 --warmup 3 --nt-kernel $k $QUIET > gpurun_out/ab.json 2>/dev/null
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('$DT nt_kernel $k: %.1f samples/s  %.3f ms/step (median %.3f)' % (d['value'], d['ms_per_step'], d['ms_per_step_median']))"
done; done
if [ "$2" = stats ]; then
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --dtype $DT --batch 1024 --steps 6 --warmup 2 --nt-kernel 81 $QUIET > gpurun_out/pf.log 2>&1
  python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/nt81_stats_$DT.txt 2>&1; rm -rf gpurun_out/pf
  grep -E "gemm_nt" gpurun_out/nt81_stats_$DT.txt | cut -c1-100,118-175
fi
