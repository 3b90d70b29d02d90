export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_kernels.py tests/test_model_parity.py tests/test_bench_launch.py tests/test_lxrt.py -m gpu -q --tb=short -p no:cacheprovider -k "attention or nlvr2 or driver_command or lxrt" > gpurun_out/r03d_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r03d_pytest.log
timeout 120 python tools/attn_bench.py 64 416 > gpurun_out/r03d_attn_s416.txt 2>&1
tail -n 12 gpurun_out/r03d_pytest.log; cat gpurun_out/r03d_attn_s416.txt
python - <<'PY'
import json
d = json.load(open("gpurun_out/parity_small.json"))
print({k: v for k, v in d.items() if "lxrt_encoder" in k})
PY
