# usage: bash tools/gpu_pmc.sh <tag> [batch] -- bench + kernel trace + PMC passes (separate runs, kernel-trace only)
TAG=${1:-pmc}; B=${2:-64}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python bench.py --steps 10 --warmup 3 --batch $B --no-cpu-baseline > gpurun_out/${TAG}_bench_b$B.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_bench_b$B.log
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_trace -o t -- python bench.py --steps 3 --warmup 2 --batch $B --no-cpu-baseline --no-profile > gpurun_out/${TAG}_trace.log 2>&1
python tools/rocpd_summary.py gpurun_out/${TAG}_trace/t_results.db --by-grid > gpurun_out/${TAG}_kernel_stats_by_grid.txt 2>&1
python tools/rocpd_summary.py gpurun_out/${TAG}_trace/t_results.db > gpurun_out/${TAG}_kernel_stats.txt 2>&1
rm -rf gpurun_out/${TAG}_trace
rocprofv3 -L > gpurun_out/${TAG}_counters_list.txt 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/${TAG}_pmc1 -o p -- python bench.py --steps 1 --warmup 1 --batch $B --no-cpu-baseline --no-profile > gpurun_out/${TAG}_pmc1.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_pmc1.log
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d gpurun_out/${TAG}_pmc2 -o p -- python bench.py --steps 1 --warmup 1 --batch $B --no-cpu-baseline --no-profile > gpurun_out/${TAG}_pmc2.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_pmc2.log
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/${TAG}_pmc3 -o p -- python bench.py --steps 1 --warmup 1 --batch $B --no-cpu-baseline --no-profile > gpurun_out/${TAG}_pmc3.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_pmc3.log
python tools/rocpd_pmc.py gpurun_out/${TAG}_pmc1/p_results.db > gpurun_out/${TAG}_pmc1.txt 2>&1
python tools/rocpd_pmc.py gpurun_out/${TAG}_pmc2/p_results.db > gpurun_out/${TAG}_pmc2.txt 2>&1
python tools/rocpd_pmc.py gpurun_out/${TAG}_pmc3/p_results.db > gpurun_out/${TAG}_pmc3.txt 2>&1
ls -la gpurun_out/${TAG}_pmc1 gpurun_out/${TAG}_pmc2
rm -rf gpurun_out/${TAG}_pmc2 gpurun_out/${TAG}_pmc3
grep -h '^{' gpurun_out/${TAG}_bench_b$B.log | cut -c1-1600
head -16 gpurun_out/${TAG}_kernel_stats.txt | cut -c1-190
head -30 gpurun_out/${TAG}_pmc1.txt | cut -c1-220
