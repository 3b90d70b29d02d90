# round-4 session 14 (last): the other two BASELINE workloads on the final tree + the headline at per-GPU batch 1536 (measurement only)
export TMPDIR=/tmp
mkdir -p gpurun_out
QUIET="--no-cpu-baseline --strict-dtype none --no-vendor-leg --pmc-traffic off"
timeout 200 python bench.py --workload vqa --steps 20 --warmup 5 $QUIET > gpurun_out/r04_final5_bench_vqa.json 2> gpurun_out/vqa.err
timeout 200 python bench.py --workload nlvr2 --steps 20 --warmup 5 $QUIET > gpurun_out/r04_final5_bench_nlvr2.json 2> gpurun_out/nlvr2.err
timeout 200 python bench.py --batch 1536 --steps 12 --warmup 4 $QUIET --no-profile --no-h2d --no-parity > gpurun_out/r04_final5_bench_b1536.json 2> gpurun_out/b1536.err
timeout 200 python bench.py --dtype bf16x3 --batch 512 --steps 12 --warmup 4 $QUIET --no-profile --no-h2d --no-parity > gpurun_out/r04_final5_bench_x3_b512.json 2> gpurun_out/x3b512.err
python - <<PY
import json
for w in ("vqa","nlvr2","b1536","x3_b512"):
    try:
        e=json.loads([l for l in open("gpurun_out/r04_final5_bench_%s.json"%w) if l.startswith("{")][-1]); print(w, e["value"], e["ms_per_step"], e["config"].get("per_gpu_batch"))
    except Exception as ex: print(w, "failed", ex)
PY
tail -2 gpurun_out/b1536.err
