# round-3 GPU session 3: why the bf16x3 step is host-bound (plain run, HIP-API trace), fp32-kernel step for comparison, default bench, 2-rank launch test
export TMPDIR=/tmp
mkdir -p gpurun_out
X3="--batch 128 --steps 4 --warmup 2 --no-cpu-baseline --no-h2d --no-profile --no-parity --strict-dtype none"
timeout 120 python bench.py --dtype bf16x3 $X3 > gpurun_out/r03c_bench_x3_plain.json 2> gpurun_out/r03c_x3_plain.err
timeout 200 rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d gpurun_out/ph -o h -- python bench.py --dtype bf16x3 $X3 > gpurun_out/r03c_bench_x3_hiptrace.json 2> gpurun_out/ph.log
for f in $(find gpurun_out/ph -name "*hip_api_stats.csv" | head -1); do head -25 $f > gpurun_out/r03c_x3_hip_api_stats.csv; done
for f in $(find gpurun_out/ph -name "*kernel_stats.csv" | head -1); do head -25 $f > gpurun_out/r03c_x3_kernel_stats.csv; done
ls -R gpurun_out/ph | head -20 > gpurun_out/r03c_ph_ls.txt; rm -rf gpurun_out/ph
timeout 150 python bench.py --dtype fp32 --batch 128 --steps 3 --warmup 1 --no-cpu-baseline --no-h2d --no-profile --no-parity --strict-dtype none > gpurun_out/r03c_bench_fp32_b128.json 2> gpurun_out/r03c_fp32.err
timeout 400 python bench.py > gpurun_out/r03c_bench.json 2> gpurun_out/r03c_bench.err; echo "rc=$?" >> gpurun_out/r03c_bench.err
timeout 280 python -m pytest tests/test_bench_launch.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r03c_pytest_launch.log 2>&1; echo "rc=$?" >> gpurun_out/r03c_pytest_launch.log
cut -c1-700 gpurun_out/r03c_bench_x3_plain.json; echo; cut -c1-300 gpurun_out/r03c_bench_x3_hiptrace.json; echo
cat gpurun_out/r03c_x3_hip_api_stats.csv; cat gpurun_out/r03c_ph_ls.txt
cut -c1-500 gpurun_out/r03c_bench_fp32_b128.json; echo
cut -c1-4000 gpurun_out/r03c_bench.json; tail -3 gpurun_out/r03c_bench.err
tail -n 8 gpurun_out/r03c_pytest_launch.log
