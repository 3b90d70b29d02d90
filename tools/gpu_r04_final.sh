# round-4 "state of the tree" GPU session: whole GPU suite, smoke, the default bench line (every leg, in-run PMC traffic for both
# timed modes).  Kernel-stats summaries of the two timed commands: tools/gpu_r04_run4.sh (same kernels).   usage: bash tools/gpu_r04_final.sh <tag>
TAG=${1:-r04_final}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1100 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_smoke.log
START=$(date +%s)
timeout 1200 python bench.py > gpurun_out/${TAG}_bench_b1024.json 2> gpurun_out/${TAG}_bench.err; echo "rc=$? wall=$(( $(date +%s) - START ))s" >> gpurun_out/${TAG}_bench.err
START=$(date +%s)
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_form.json 2>> gpurun_out/${TAG}_bench.err; echo "driver form rc=$? wall=$(( $(date +%s) - START ))s" >> gpurun_out/${TAG}_bench.err
tail -n 6 gpurun_out/${TAG}_pytest.log; tail -n 3 gpurun_out/${TAG}_smoke.log
python - <<PY
import json
for f in ("b1024", "driver_form"):
    d=json.load(open("gpurun_out/${TAG}_bench_%s.json" % f))
    r=d["roofline"]; s=d["strict_mode"]
    print(f, "bf16", d["value"], d["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"], r["traffic_source"][:60])
    print(f, "x3", s["value"], s["ms_per_step"], s["steps"], s["max_dlogit"], "frac", s["roofline"]["frac"], "traffic", s["roofline"]["traffic"], s["roofline"]["traffic_source"][:200])
    print(f, "fp32", s["fp32_kernels"]["value"], "vendor", d["vendor_plain_gemms"]["value"], "cpu", d["cpu_baseline"]["value"])
PY
tail -3 gpurun_out/${TAG}_bench.err
