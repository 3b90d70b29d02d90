# round-4 GPU session 10: with streaming stores in, which NT kernel wins INSIDE the bf16 step?  auto (k90 except plain K >= 2048) vs 81 vs 90, alternating
TAG=${1:-r04_run10}
export TMPDIR=/tmp
mkdir -p gpurun_out
COMMON="--steps 30 --warmup 5 --no-cpu-baseline --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off"
for rep in 1 2; do
  for k in 0 81 90; do
    timeout 200 python bench.py $COMMON --nt-kernel $k > gpurun_out/${TAG}_k${k}_$rep.json 2>> gpurun_out/${TAG}.err
  done
done
python - <<PY
import json
for k in (0,81,90):
    for rep in (1,2):
        d=json.load(open("gpurun_out/${TAG}_k%d_%d.json"%(k,rep))); r=d["roofline"]
        print("nt_kernel", k, "rep", rep, d["value"], d["ms_per_step"], "gemm ms/step", r["gemm_ms_per_step"], {kk[:20]:v["ms_per_step"] for kk,v in r["by_kernel"].items()})
PY
