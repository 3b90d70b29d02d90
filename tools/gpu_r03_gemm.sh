export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 200 python tools/gemm_ab.py ${1:-512} 90 100 101 > gpurun_out/r03_gemm_ab_k101.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03_gemm_ab_k101.txt
