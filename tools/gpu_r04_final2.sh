# round-4 final "state of the tree" session (after the streaming-store change): whole GPU suite, smoke, default bench line + driver form,
# rocprofv3 kernel stats of the two timed commands, NT shapes alone.
TAG=${1:-r04_final}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1100 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_smoke.log
START=$(date +%s)
timeout 1200 python bench.py > gpurun_out/${TAG}_bench_b1024.json 2> gpurun_out/${TAG}_bench.err; echo "rc=$? wall=$(( $(date +%s) - START ))s" >> gpurun_out/${TAG}_bench.err
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_form.json 2>> gpurun_out/${TAG}_bench.err
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off"
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --steps 8 --warmup 2 $QUIET > gpurun_out/pf.log 2>&1
python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/${TAG}_kernel_stats_b1024.txt 2>&1; rm -rf gpurun_out/pf
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --dtype bf16x3 --batch 1024 --steps 8 --warmup 2 $QUIET > gpurun_out/pf2.log 2>&1
python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/${TAG}_kernel_stats_bf16x3_b1024.txt 2>&1; rm -rf gpurun_out/pf
VB_DEV=1 VB_NOCHECK=1 timeout 300 python tools/gemm_ab.py 1024 90 81 > gpurun_out/${TAG}_gemm_ab.txt 2>&1
tail -n 6 gpurun_out/${TAG}_pytest.log; tail -n 3 gpurun_out/${TAG}_smoke.log
python - <<PY
import json
for f in ("b1024", "driver_form"):
    d=json.load(open("gpurun_out/${TAG}_bench_%s.json" % f))
    r=d["roofline"]; s=d["strict_mode"]
    print(f, "bf16", d["value"], d["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"], r["traffic_source"][-60:])
    print(f, "x3", s["value"], s["ms_per_step"], s["steps"], s["max_dlogit"], "frac", s["roofline"]["frac"], "traffic", s["roofline"]["traffic"])
    print(f, "fp32", s["fp32_kernels"]["value"], "vendor", d["vendor_plain_gemms"]["value"], "cpu", d["cpu_baseline"]["value"])
PY
tail -2 gpurun_out/${TAG}_bench.err; cut -c1-130 gpurun_out/${TAG}_gemm_ab.txt
