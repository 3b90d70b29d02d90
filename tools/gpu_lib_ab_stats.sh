# per-kernel A/B of two product builds under the kernel trace, both timed modes: $1 = base library, $2 = candidate library
export TMPDIR=/tmp
mkdir -p gpurun_out
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off"
for mode in bf16 bf16x3; do for lib in $1 $2 $1 $2; do
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --dtype $mode --batch 1024 --steps 6 --warmup 2 --lib-path $lib $QUIET > gpurun_out/pf.log 2>&1
  python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/abk.txt 2>&1; rm -rf gpurun_out/pf
  echo "$mode $(basename $lib): $(head -1 gpurun_out/abk.txt | cut -c1-60)"; grep -E "gemm_nt_dual" gpurun_out/abk.txt | awk '{printf "    %-70s %6s %10s\n", substr($0,1,70), $(NF-5), $(NF-3)}' | head -6
done; done
