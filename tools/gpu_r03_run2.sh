# round-3 GPU session 2: whole GPU suite (incl. bf16x3 + bench-shape tests), default bench with strict modes, vendor kernel names,
# kernel breakdown of the bf16x3 step
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --deselect tests/test_bench_launch.py::test_driver_command_two_ranks_end_to_end > gpurun_out/r03b_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r03b_pytest.log
timeout 280 python -m pytest tests/test_bench_launch.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r03b_pytest_launch.log 2>&1; echo "rc=$?" >> gpurun_out/r03b_pytest_launch.log
timeout 400 python bench.py > gpurun_out/r03b_bench.json 2> gpurun_out/r03b_bench.err; echo "rc=$?" >> gpurun_out/r03b_bench.err
timeout 120 rocprofv3 --kernel-trace --stats -d gpurun_out/pv -o v -- python tools/vendor_kernel_names.py 1024 > gpurun_out/pv.log 2>&1
python tools/rocpd_summary.py gpurun_out/pv/v_results.db 2>&1 | cut -c1-400 > gpurun_out/r03_vendor_kernel_names.txt; rm -rf gpurun_out/pv
timeout 200 rocprofv3 --kernel-trace --stats -d gpurun_out/px -o x -- python bench.py --dtype bf16x3 --batch 128 --steps 4 --warmup 2 --no-cpu-baseline --no-h2d --no-profile --no-parity > gpurun_out/r03b_bench_x3_b128.json 2> gpurun_out/px.log
python tools/rocpd_summary.py gpurun_out/px/x_results.db 2>&1 | cut -c1-200 > gpurun_out/r03b_kernel_stats_x3_b128.txt; rm -rf gpurun_out/px
tail -n 15 gpurun_out/r03b_pytest.log; tail -n 12 gpurun_out/r03b_pytest_launch.log
cut -c1-3000 gpurun_out/r03b_bench.json; tail -3 gpurun_out/r03b_bench.err
head -30 gpurun_out/r03_vendor_kernel_names.txt
cut -c1-800 gpurun_out/r03b_bench_x3_b128.json; head -30 gpurun_out/r03b_kernel_stats_x3_b128.txt
