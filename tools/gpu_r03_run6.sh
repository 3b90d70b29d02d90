export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_kernels.py -m gpu -q --tb=short -p no:cacheprovider -k "attention_fwd_bwd or direct_b" > gpurun_out/r03f_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r03f_pytest.log
tail -4 gpurun_out/r03f_pytest.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/parity_small.json"))
for k in ("attention_kernel_bf16", "attention_kernel_fp32"):
    v = d.get(k, {})
    print(k, "max ctx_err", max(x["ctx_err"] for x in v.values()), "max dqkv", max(x["dqkv_err_over_gmax"] for x in v.values()))
    for name, x in sorted(v.items()): print("  ", name, x)
PY
