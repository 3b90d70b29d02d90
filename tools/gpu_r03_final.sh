# round-3 "final state" GPU session: whole GPU suite, smoke, default bench, rocprofv3 kernel stats of the same command (fewer steps),
# the other two BASELINE workloads.   usage: bash tools/gpu_r03_final.sh <tag>
TAG=${1:-r03_final}
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_smoke.log
timeout 400 python bench.py > gpurun_out/${TAG}_bench_b1024.json 2> gpurun_out/${TAG}_bench.err; echo "rc=$?" >> gpurun_out/${TAG}_bench.err
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg > gpurun_out/pf.log 2>&1
python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/${TAG}_kernel_stats_b1024.txt 2>&1; rm -rf gpurun_out/pf
timeout 200 python bench.py --workload vqa --steps 20 --warmup 5 --no-cpu-baseline --strict-dtype none > gpurun_out/${TAG}_bench_vqa.json 2> gpurun_out/vqa.err
timeout 200 python bench.py --workload nlvr2 --steps 20 --warmup 5 --no-cpu-baseline --strict-dtype none > gpurun_out/${TAG}_bench_nlvr2.json 2> gpurun_out/nlvr2.err
tail -n 6 gpurun_out/${TAG}_pytest.log; tail -n 4 gpurun_out/${TAG}_smoke.log
cut -c1-400 gpurun_out/${TAG}_bench_b1024.json; echo; tail -2 gpurun_out/${TAG}_bench.err
head -22 gpurun_out/${TAG}_kernel_stats_b1024.txt | cut -c1-190
cut -c1-260 gpurun_out/${TAG}_bench_vqa.json; echo; cut -c1-260 gpurun_out/${TAG}_bench_nlvr2.json; echo
