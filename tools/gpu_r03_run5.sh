export TMPDIR=/tmp
mkdir -p gpurun_out
X3="--steps 4 --warmup 2 --no-cpu-baseline --no-h2d --no-profile --no-parity --strict-dtype none"
timeout 200 python -m pytest tests/test_bf16x3.py tests/test_lxrt.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/r03e_pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r03e_pytest.log
for b in 128 256 512; do timeout 120 python bench.py --dtype bf16x3 --batch $b $X3 > gpurun_out/r03e_bench_x3_b$b.json 2> gpurun_out/r03e_x3_b$b.err; done
tail -n 4 gpurun_out/r03e_pytest.log
for b in 128 256 512; do cut -c1-220 gpurun_out/r03e_bench_x3_b$b.json; echo; tail -2 gpurun_out/r03e_x3_b$b.err; done
