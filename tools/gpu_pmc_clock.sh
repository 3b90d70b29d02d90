# effective shader clock while the persistent NT GEMM runs: GRBM_GUI_ACTIVE (busy cycles per XCD, summed over 8) against the
# kernel's duration from the same rocprofv3 run, full chip vs 64 workgroups, and the pure-MFMA loop for comparison.
# usage: bash tools/gpu_pmc_clock.sh   (inside a gpurun call; writes gpurun_out/pmc_clock.txt)
export TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/pmc_clock.txt
for wgs in 0 64; do
  M_ROWS=83968 WGS=$wgs rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d gpurun_out/pc -o p -- python tools/gemm_one.py 768 3072 81 > gpurun_out/pc.log 2>&1
  echo "== FFN-out shape (N=768 K=3072, M=83968), workgroups=${wgs/#0/all}" >> gpurun_out/pmc_clock.txt
  python tools/rocpd_pmc.py gpurun_out/pc/p_results.db 2>&1 | grep -E "kernel |gemm_nt_8ph" | cut -c1-60,93- >> gpurun_out/pmc_clock.txt
  python tools/rocpd_summary.py gpurun_out/pc/p_results.db 2>&1 | grep -E "gemm_nt_8ph" | cut -c1-60,119- >> gpurun_out/pmc_clock.txt
  rm -rf gpurun_out/pc
done
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d gpurun_out/pc -o p -- python tools/mfma_peak.py > gpurun_out/pc.log 2>&1
echo "== pure MFMA loop (tools/mfma_peak.py; per block-count rows in launch order)" >> gpurun_out/pmc_clock.txt
python tools/rocpd_pmc.py gpurun_out/pc/p_results.db 2>&1 | grep -E "kernel |mfma_peak" | cut -c1-60,93- >> gpurun_out/pmc_clock.txt
python tools/rocpd_summary.py gpurun_out/pc/p_results.db --by-grid 2>&1 | grep -E "mfma_peak" | cut -c1-80,119- >> gpurun_out/pmc_clock.txt
rm -rf gpurun_out/pc
cat gpurun_out/pmc_clock.txt
