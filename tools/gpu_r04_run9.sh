# round-4 GPU session 9: streaming (nt) stores in the GEMM epilogues -- every NT shape alone (k90 / k81), the step in both modes, GEMM tests
TAG=${1:-r04_run9}
export TMPDIR=/tmp
mkdir -p gpurun_out
VB_DEV=1 VB_NOCHECK=1 timeout 400 python tools/gemm_ab.py 1024 90 81 > gpurun_out/${TAG}_gemm_ab.txt 2>&1
COMMON="--steps 30 --warmup 5 --no-cpu-baseline --no-h2d --strict-dtype none --no-vendor-leg --pmc-traffic off"
timeout 300 python bench.py $COMMON > gpurun_out/${TAG}_bench_bf16.json 2> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py $COMMON --dtype bf16x3 --batch 1024 --steps 12 > gpurun_out/${TAG}_bench_x3.json 2>> gpurun_out/${TAG}_bench.err
timeout 600 python -m pytest tests/test_kernels.py tests/test_bf16x3.py tests/test_bench_shape.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm or bench_shape or layer" > gpurun_out/${TAG}_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_pytest.log
cut -c1-140 gpurun_out/${TAG}_gemm_ab.txt
python - <<PY
import json
for f in ("bf16","x3"):
    d=json.load(open("gpurun_out/${TAG}_bench_%s.json"%f)); r=d["roofline"]
    print(f, d["value"], d["ms_per_step"], "frac", r["frac"], r["avg_launch_us"], {k[:22]:v["ms_per_step"] for k,v in r["by_kernel"].items()})
PY
tail -n 5 gpurun_out/${TAG}_pytest.log
