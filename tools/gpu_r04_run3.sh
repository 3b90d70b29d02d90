# round-4 GPU session 3: split-operand step after (a) the GEMM dispatch by epilogue weight, (b) image-only results (no fp32 ctx / dfo /
# dao / dqkv in the layer).  Targeted tests, the step at B = 512 / 1024, kernel stats.
TAG=${1:-r04_run3}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bf16x3.py tests/test_bench_shape.py tests/test_parity_at_scale.py tests/test_lxrt.py tests/test_kernels.py tests/test_model_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "x3 or X3 or attention or layernorm or bench_shape or lxrt or fp32 or strict" > gpurun_out/${TAG}_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_pytest.log
COMMON="--dtype bf16x3 --steps 12 --warmup 3 --no-cpu-baseline --no-h2d --strict-dtype none --no-vendor-leg"
timeout 200 python bench.py $COMMON --batch 512 > gpurun_out/${TAG}_x3_b512.json 2> gpurun_out/${TAG}_x3.err
timeout 200 python bench.py $COMMON --batch 1024 > gpurun_out/${TAG}_x3_b1024.json 2>> gpurun_out/${TAG}_x3.err
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py $COMMON --batch 512 --steps 8 --warmup 2 --no-profile --no-parity > gpurun_out/pf.log 2>&1
python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/${TAG}_kernel_stats_bf16x3_b512.txt 2>&1; rm -rf gpurun_out/pf
tail -n 12 gpurun_out/${TAG}_pytest.log
for f in b512 b1024; do python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_x3_$f.json")); print("$f", d["value"], d["ms_per_step"], d["parity"]["max_dlogit_vs_fp32_ref"], {k[:24]:v["ms_per_step"] for k,v in d["roofline"]["by_kernel"].items()})
PY
done
tail -3 gpurun_out/${TAG}_x3.err
head -22 gpurun_out/${TAG}_kernel_stats_bf16x3_b512.txt | cut -c1-170
