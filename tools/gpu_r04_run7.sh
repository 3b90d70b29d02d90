# round-4 GPU session 7: polynomial GELU / GELU' in the bf16 FFN-in epilogue -- bf16 parity tests, the shape alone, the step
TAG=${1:-r04_run7}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels.py tests/test_bench_shape.py tests/test_model_parity.py tests/test_parity_at_scale.py -m gpu -q --tb=short -p no:cacheprovider -k "not x3 and not fp32" > gpurun_out/${TAG}_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_pytest.log
VB_DEV=1 VB_NOCHECK=1 timeout 300 python tools/gemm_ab.py 1024 90 81 > gpurun_out/${TAG}_gemm_ab.txt 2>&1
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-h2d --strict-dtype none --no-vendor-leg --pmc-traffic off > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
tail -n 8 gpurun_out/${TAG}_pytest.log
grep -E "gelu|per step" gpurun_out/${TAG}_gemm_ab.txt
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench.json")); r=d["roofline"]
print("bf16", d["value"], d["ms_per_step"], "frac", r["frac"], r["avg_launch_us"], d["parity"]["max_dlogit_vs_fp32_ref"], d["parity"]["mean"])
PY
python - <<PY
import json
d=json.load(open("gpurun_out/parity_at_scale.json"))
b=d["base_pretraining_b16"]["bf16"]; print({k:b[k] for k in ("max_dlogit","mean_dlogit","grad_rel_l2_median","grad_rel_l2_worst","top1_agree")})
PY
