# round-3 GPU session 1: new bench-shape tests, vendor yardstick, attention in-step experiment, per-kernel clocks, GEMM PMC, default bench
export TMPDIR=/tmp
mkdir -p gpurun_out
rocminfo | grep -E "Marketing Name|gfx" | head -2 > gpurun_out/r03a_gpu.txt 2>&1
timeout 900 python -m pytest tests/test_bench_shape.py tests/test_bench_launch.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r03a_pytest_new.log 2>&1; echo "rc=$?" >> gpurun_out/r03a_pytest_new.log
timeout 400 python tools/gemm_vendor_yardstick.py 512 1024 > gpurun_out/r03_gemm_vendor_yardstick.txt 2>&1
(timeout 200 python tools/attn_instep.py 512; timeout 200 python tools/attn_instep.py 1024) > gpurun_out/r03_attn_instep.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d gpurun_out/pclk -o p -- python bench.py --steps 2 --warmup 1 --batch 512 --no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none > gpurun_out/pclk.log 2>&1
python tools/rocpd_clock.py gpurun_out/pclk/p_results.db > gpurun_out/r03_clock_in_step_b512.txt 2>&1; rm -rf gpurun_out/pclk
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d gpurun_out/pclk -o p -- python tools/attn_bench.py 512 > gpurun_out/pclk2.log 2>&1
python tools/rocpd_clock.py gpurun_out/pclk/p_results.db attn > gpurun_out/r03_clock_attn_alone_b512.txt 2>&1; rm -rf gpurun_out/pclk
timeout 600 bash tools/gpu_pmc_gemm_r03.sh > gpurun_out/r03_pmc_gemm_all.log 2>&1
timeout 900 python bench.py > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err; echo "rc=$?" >> gpurun_out/r03a_bench.err
tail -n 5 gpurun_out/r03a_pytest_new.log
cat gpurun_out/r03_gemm_vendor_yardstick.txt | tail -n 24
cat gpurun_out/r03_attn_instep.txt
head -20 gpurun_out/r03_clock_in_step_b512.txt; cat gpurun_out/r03_clock_attn_alone_b512.txt
cut -c1-1500 gpurun_out/r03a_bench.json
