# MFMA-pipe utilisation per kernel inside the training step, both timed modes (two PMC passes each, kernel trace only)
export TMPDIR=/tmp
mkdir -p gpurun_out
QUIET="--steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off"
SETA="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
SETB="GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_INST_LEVEL_VMEM"
for mode in bf16 bf16x3; do
  timeout 400 rocprofv3 --kernel-trace --pmc $SETA -d gpurun_out/pa_$mode -o p -- python bench.py --dtype $mode --batch 1024 $QUIET > gpurun_out/pa_$mode.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc $SETB -d gpurun_out/pb_$mode -o p -- python bench.py --dtype $mode --batch 1024 $QUIET > gpurun_out/pb_$mode.log 2>&1
  A=$(find gpurun_out/pa_$mode -name "*_results.db" | head -1); B=$(find gpurun_out/pb_$mode -name "*_results.db" | head -1)
  echo "# python bench.py --dtype $mode --batch 1024 (3 steps), rocprofv3 --kernel-trace --pmc, two passes; tools/pmc_step_summary.py" > gpurun_out/r04_pmc_step_$mode.txt
  python tools/pmc_step_summary.py $A $B >> gpurun_out/r04_pmc_step_$mode.txt 2>&1
  rm -rf gpurun_out/pa_$mode gpurun_out/pb_$mode
  cat gpurun_out/r04_pmc_step_$mode.txt
done
