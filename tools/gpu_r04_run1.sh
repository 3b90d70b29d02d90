# round-4 GPU session 1: validate the round's plumbing (config row, dev-library split, bench restructure, bf16x3 bench-shape tests,
# specialised split-operand epilogues) and collect the baselines the kernel work starts from.   usage: bash tools/gpu_r04_run1.sh <tag>
TAG=${1:-r04_run1}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1100 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_smoke.log
timeout 500 python bench.py > gpurun_out/${TAG}_bench_b1024.json 2> gpurun_out/${TAG}_bench.err; echo "rc=$?" >> gpurun_out/${TAG}_bench.err
timeout 300 python bench.py --dtype bf16x3 --batch 1024 --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --strict-dtype none --no-vendor-leg > gpurun_out/${TAG}_bench_x3_b1024.json 2> gpurun_out/${TAG}_bench_x3_b1024.err
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --dtype bf16x3 --batch 512 --steps 8 --warmup 2 --no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg > gpurun_out/pf.log 2>&1
python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/${TAG}_kernel_stats_bf16x3_b512.txt 2>&1; rm -rf gpurun_out/pf
timeout 300 python tools/power_by_kernel.py > gpurun_out/${TAG}_power_by_kernel.txt 2> gpurun_out/${TAG}_power.err
tail -n 8 gpurun_out/${TAG}_pytest.log; tail -n 4 gpurun_out/${TAG}_smoke.log
cut -c1-300 gpurun_out/${TAG}_bench_b1024.json; echo; tail -2 gpurun_out/${TAG}_bench.err
cut -c1-300 gpurun_out/${TAG}_bench_x3_b1024.json; echo; tail -2 gpurun_out/${TAG}_bench_x3_b1024.err
head -12 gpurun_out/${TAG}_kernel_stats_bf16x3_b512.txt | cut -c1-170
cat gpurun_out/${TAG}_power_by_kernel.txt; tail -3 gpurun_out/${TAG}_power.err
