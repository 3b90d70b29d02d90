# round-4 GPU session 5: the default bench line (every leg) + the two other BASELINE workloads
TAG=${1:-r04_run5}
export TMPDIR=/tmp
mkdir -p gpurun_out
START=$(date +%s)
timeout 1200 python bench.py > gpurun_out/${TAG}_bench_b1024.json 2> gpurun_out/${TAG}_bench.err; echo "rc=$? wall=$(( $(date +%s) - START ))s" >> gpurun_out/${TAG}_bench.err
QUIET="--no-cpu-baseline --strict-dtype none --no-vendor-leg --pmc-traffic off"
timeout 300 python bench.py --workload vqa --steps 20 --warmup 5 $QUIET > gpurun_out/${TAG}_bench_vqa.json 2> gpurun_out/vqa.err
timeout 300 python bench.py --workload nlvr2 --steps 20 --warmup 5 $QUIET > gpurun_out/${TAG}_bench_nlvr2.json 2> gpurun_out/nlvr2.err
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench_b1024.json"))
r=d["roofline"]; s=d["strict_mode"]
print("bf16", d["value"], d["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"], r["traffic_source"][:230])
print("x3", s["value"], s["ms_per_step"], s["max_dlogit"], "frac", s["roofline"]["frac"], "traffic", s["roofline"]["traffic"], s["roofline"]["traffic_source"][:230])
print("fp32", s["fp32_kernels"]["value"], "vendor", d["vendor_plain_gemms"]["value"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"][-60:])
for w in ("vqa","nlvr2"):
    e=json.load(open("gpurun_out/${TAG}_bench_%s.json"%w)); print(w, e["value"], e["ms_per_step"])
PY
tail -3 gpurun_out/${TAG}_bench.err
