# in-step A/B of two PRODUCT builds on one box: $1 = the reference build (e.g. tools/libvisualbert_hip_ab_old.so), the in-tree library is the candidate;
# plain timing (3 interleaved rounds) + one kernel trace each
export TMPDIR=/tmp
mkdir -p gpurun_out
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
OLD=${1:?usage: gpu_lib_ab.sh <reference build of libvisualbert_hip.so>}
NEW=visualbert_amd/libvisualbert_hip.so
for r in 1 2 3; do for lib in $OLD $NEW; do
  timeout 300 python bench.py --steps 15 --warmup 4 --lib-path $lib $QUIET > gpurun_out/ab.json 2>/dev/null
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('$lib: %.1f samples/s  %.3f ms/step (median %.3f)' % (d['value'], d['ms_per_step'], d['ms_per_step_median']))"
done; done
for lib in $OLD $NEW; do
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --steps 8 --warmup 2 --lib-path $lib $QUIET > gpurun_out/pf.log 2>&1
  python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/ab_stats_$(basename $lib .so).txt 2>&1; rm -rf gpurun_out/pf
  echo "$lib: $(head -1 gpurun_out/ab_stats_$(basename $lib .so).txt)"; grep -E "gemm_nt" gpurun_out/ab_stats_$(basename $lib .so).txt | awk '{printf "  %-90s %6s %10s\n", substr($1,1,90), $(NF-5), $(NF-3)}' | head -9
done
