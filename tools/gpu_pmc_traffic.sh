# HBM traffic of the dominant GEMM kernel (bench.py's roofline.traffic): two separate PMC passes over the bench
# command (FETCH_SIZE, WRITE_SIZE; kernel-trace only -- gpurun refuses PMC together with API tracing), reduced by
# tools/pmc_traffic.py into profiles/pmc_traffic.json.   usage: bash tools/gpu_pmc_traffic.sh [batch]
export TMPDIR=/tmp
B=${1:-128}
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pt_$c -o p -- python bench.py --steps 2 --warmup 1 --batch $B --no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg > gpurun_out/pt_$c.log 2>&1
  python tools/rocpd_pmc.py gpurun_out/pt_$c/p_results.db > gpurun_out/pt_$c.txt 2>&1
  rm -rf gpurun_out/pt_$c
done
python tools/pmc_traffic.py gpurun_out/pt_FETCH_SIZE.txt gpurun_out/pt_WRITE_SIZE.txt $B > gpurun_out/pmc_traffic.json
cat gpurun_out/pmc_traffic.json
