# round-4 session 13: LayerNorm backward with 12 columns per lane (H = 768) and the "rebuild x-hat from y" forward (3 tensors instead of 4):
# A/B of the step with rocprofv3 kernel stats per configuration + the LayerNorm / layer parity tests.
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels.py tests/test_bench_shape.py -m gpu -q --tb=short -p no:cacheprovider -k "layernorm or guard or (bench and bf16 and not x3 and not dropout and not logits)" > gpurun_out/r13_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r13_pytest.log
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off"
run() {  # tag, env...
  tag=$1; shift
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --steps 8 --warmup 2 $QUIET > gpurun_out/r13_$tag.json 2> gpurun_out/r13_$tag.err
  python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/r13_stats_$tag.txt 2>&1; rm -rf gpurun_out/pf
  python - <<PY
import json,re
try:
    d=json.loads([l for l in open("gpurun_out/r13_$tag.json") if l.startswith("{")][-1]); v=(d["value"], d["ms_per_step"])
except Exception as e: v=("?", str(e))
rows=[l for l in open("gpurun_out/r13_stats_$tag.txt") if re.search(r"ln_fwd|ln_bwd", l)]
print("$tag", v, [(re.split(r"\s{2,}", r.strip())[0][:44], re.split(r"\s{2,}", r.strip())[3]) for r in rows])
PY
}
run new_rebuild VB_X=0
run new_z VB_LN_NOREBUILD=1
run old_z VB_LN_EXP=1 VB_LN_NOREBUILD=1
run old_rebuild VB_LN_EXP=1
run old3_rebuild VB_LN_EXP=3
tail -n 4 gpurun_out/r13_pytest.log
