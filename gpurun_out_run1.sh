mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/gpu.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 300 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 600 python bench.py --steps 5 --warmup 2 --batch 16 --no-cpu-baseline > gpurun_out/bench_b16.log 2>&1; echo "rc=$?" >> gpurun_out/bench_b16.log
timeout 900 python bench.py --steps 10 --warmup 3 --batch 64 > gpurun_out/bench_b64.log 2>&1; echo "rc=$?" >> gpurun_out/bench_b64.log
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_r1 -o bench -- python bench.py --steps 4 --warmup 2 --batch 64 --no-cpu-baseline --no-profile > gpurun_out/rocprof.log 2>&1; echo "rc=$?" >> gpurun_out/rocprof.log
tail -n 6 gpurun_out/pytest_gpu.log gpurun_out/smoke.log gpurun_out/bench_b16.log gpurun_out/bench_b64.log gpurun_out/rocprof.log
ls -la gpurun_out/prof_r1 2>/dev/null | head
