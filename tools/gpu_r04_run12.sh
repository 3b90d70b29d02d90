# round-4 GPU session 12: the three plane-pair passes of the split-operand weight gradient as ONE grouped launch (12 problems per layer)
TAG=${1:-r04_run12}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_bf16x3.py tests/test_bench_shape.py tests/test_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "x3 or wgrad" > gpurun_out/${TAG}_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_pytest.log
COMMON="--dtype bf16x3 --batch 1024 --steps 12 --warmup 3 --no-cpu-baseline --no-h2d --strict-dtype none --no-vendor-leg --pmc-traffic off"
timeout 300 python bench.py $COMMON > gpurun_out/${TAG}_x3_b1024.json 2> gpurun_out/${TAG}.err
timeout 300 python bench.py $COMMON --batch 512 > gpurun_out/${TAG}_x3_b512.json 2>> gpurun_out/${TAG}.err
tail -n 4 gpurun_out/${TAG}_pytest.log
python - <<PY
import json
for f in ("b1024","b512"):
    d=json.load(open("gpurun_out/${TAG}_x3_%s.json"%f)); r=d["roofline"]
    print(f, d["value"], d["ms_per_step"], d["parity"]["max_dlogit_vs_fp32_ref"], {k.split('<')[0]+('x3' if 'x3' in k else ''):(v["ms_per_step"], v["launches_per_step"]) for k,v in r["by_kernel"].items()})
PY
