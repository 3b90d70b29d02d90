# usage: bash tools/gpu_run.sh <tag> [batch]   -- GPU-box session: parity tests, smoke, bench, rocprofv3 kernel stats
TAG=${1:-run}; B=${2:-64}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 300 > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/${TAG}_smoke.log
timeout 900 python bench.py --steps 10 --warmup 3 --batch $B > gpurun_out/${TAG}_bench_b$B.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_bench_b$B.log
timeout 600 python bench.py --steps 10 --warmup 3 --batch 128 --no-cpu-baseline > gpurun_out/${TAG}_bench_b128.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_bench_b128.log
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_prof -o bench -- python bench.py --steps 4 --warmup 2 --batch $B --no-cpu-baseline --no-profile > gpurun_out/${TAG}_rocprof.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_rocprof.log
python tools/rocpd_summary.py gpurun_out/${TAG}_prof/bench_results.db > gpurun_out/${TAG}_kernel_stats.txt 2>&1
rm -rf gpurun_out/${TAG}_prof
tail -n 4 gpurun_out/${TAG}_pytest.log gpurun_out/${TAG}_smoke.log
grep -h '^{' gpurun_out/${TAG}_bench_b$B.log gpurun_out/${TAG}_bench_b128.log | cut -c1-1800
head -24 gpurun_out/${TAG}_kernel_stats.txt | cut -c1-200
