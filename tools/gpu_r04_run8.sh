# round-4 GPU session 8: one-byte saved gelu' in the bf16 encoder layer (VB_ACT_GELU_SAVE_GRAD8 / VB_ACT_MUL_AUX8)
TAG=${1:-r04_run8}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels.py tests/test_bench_shape.py tests/test_model_parity.py tests/test_parity_at_scale.py -m gpu -q --tb=short -p no:cacheprovider -k "not x3" > gpurun_out/${TAG}_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_pytest.log
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-h2d --strict-dtype none --no-vendor-leg --pmc-traffic off > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off"
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --steps 8 --warmup 2 $QUIET > gpurun_out/pf.log 2>&1
python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/${TAG}_kernel_stats_b1024.txt 2>&1; rm -rf gpurun_out/pf
tail -n 8 gpurun_out/${TAG}_pytest.log
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench.json")); r=d["roofline"]
print("bf16", d["value"], d["ms_per_step"], "frac", r["frac"], r["avg_launch_us"], d["parity"]["max_dlogit_vs_fp32_ref"], d["parity"]["mean"])
d=json.load(open("gpurun_out/parity_at_scale.json"))
b=d["base_pretraining_b16"]["bf16"]; print({k:b[k] for k in ("max_dlogit","mean_dlogit","grad_rel_l2_median","grad_rel_l2_worst","adam_delta_rel_l2_median","top1_agree")})
PY
head -14 gpurun_out/${TAG}_kernel_stats_b1024.txt | cut -c1-170
