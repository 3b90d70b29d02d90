# round-4 GPU session 4 ("state of the tree"): whole GPU suite, smoke, the default bench line (bf16 headline + in-run PMC traffic +
# bf16x3 strict leg with its own roofline + fp32 + vendor yardstick + CPU baseline), rocprofv3 kernel stats of both timed modes.
TAG=${1:-r04_run4}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1100 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_smoke.log
/usr/bin/time -v timeout 900 python bench.py > gpurun_out/${TAG}_bench_b1024.json 2> gpurun_out/${TAG}_bench.err; echo "rc=$?" >> gpurun_out/${TAG}_bench.err
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off"
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --steps 8 --warmup 2 $QUIET > gpurun_out/pf.log 2>&1
python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/${TAG}_kernel_stats_b1024.txt 2>&1; rm -rf gpurun_out/pf
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --dtype bf16x3 --batch 1024 --steps 8 --warmup 2 $QUIET > gpurun_out/pf2.log 2>&1
python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/${TAG}_kernel_stats_bf16x3_b1024.txt 2>&1; rm -rf gpurun_out/pf
tail -n 8 gpurun_out/${TAG}_pytest.log; tail -n 3 gpurun_out/${TAG}_smoke.log
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench_b1024.json"))
r=d["roofline"]; s=d["strict_mode"]
print("bf16", d["value"], d["ms_per_step"], "frac", r["frac"], "traffic", r["traffic"], r["traffic_source"][:160])
print("x3", s["value"], s["ms_per_step"], s["max_dlogit"], "frac", s["roofline"]["frac"], "traffic", s["roofline"]["traffic"], s["roofline"]["traffic_source"][:160])
print("fp32", s["fp32_kernels"]["value"], "vendor", d["vendor_plain_gemms"]["value"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["sample"][-40:])
PY
grep -E "Elapsed|rc=" gpurun_out/${TAG}_bench.err
head -14 gpurun_out/${TAG}_kernel_stats_b1024.txt | cut -c1-170
head -14 gpurun_out/${TAG}_kernel_stats_bf16x3_b1024.txt | cut -c1-170
