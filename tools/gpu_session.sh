#!/bin/bash
# ONE parameterised GPU session script (replaces the per-session tools/gpu_r0N_run*.sh of rounds 1-4):
#
#     gpurun --timeout 900 -- 'bash tools/gpu_session.sh TAG step [step ...]'
#
# Every step writes gpurun_out/${TAG}_<step>.* ; the summaries worth keeping are copied to profiles/ by hand afterwards.
#   gate        fp8-cross-term logits gate on the three pre-registered inputs (tools/x3f8_logits.py)
#   stress      tests/test_parity_at_scale.py -k stress (trained-like golden of the real reference, three modes)
#   pytest      the whole GPU suite
#   smoke       __graft_entry__.smoke()
#   bench       default bench.py line (the driver's command) -> ${TAG}_bench.json
#   stats       rocprofv3 --kernel-trace --stats of a short quiet bench run, bf16 and bf16x3
#   pmc         per-kernel PMC passes of the bf16 step (MFMA busy, LDS conflicts, waits; tools/gpu_pmc_step)
#   gemmab:<arms>   tools/gemm_ab.py 1024 <arms...>   (developer library; arms separated by ',')
#   attn        tools/attn_bench.py at the bench shape (B = 1024, S = 164) + in-step stats
#   nlvr2real   bench.py --workload nlvr2-real + kernel stats
#   sweep       batch sweep B = 8 / 64 / 256 / 1024, bf16 and bf16x3 (tools/batch_sweep.py)
#   py:<file>   python <file> (any extra tool; ':' separated arguments)
TAG=${1:-r05}
shift
export TMPDIR=/tmp
mkdir -p gpurun_out
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
for step in "$@"; do
  echo "=== $step ($(date +%T))"
  case "$step" in
    gate)
      timeout 600 python tools/x3f8_logits.py --cases base_pretraining_b16 base_pretraining_stress_b8 trained > gpurun_out/${TAG}_x3f8_gate.txt 2> gpurun_out/${TAG}_x3f8_gate.err
      echo "rc=$?" >> gpurun_out/${TAG}_x3f8_gate.err; cat gpurun_out/${TAG}_x3f8_gate.txt; tail -n 3 gpurun_out/${TAG}_x3f8_gate.err ;;
    stress)
      timeout 900 python -m pytest tests/test_parity_at_scale.py -m gpu -q -k stress --tb=short -p no:cacheprovider -s > gpurun_out/${TAG}_stress.log 2>&1
      echo "rc=$?" >> gpurun_out/${TAG}_stress.log; grep -E "^stress @|passed|failed|rc=|Error|assert" gpurun_out/${TAG}_stress.log | cut -c1-900 ;;
    pytest)
      timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 900 > gpurun_out/${TAG}_pytest.log 2>&1
      echo "rc=$?" >> gpurun_out/${TAG}_pytest.log; tail -n 8 gpurun_out/${TAG}_pytest.log ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1; echo "rc=$?" >> gpurun_out/${TAG}_smoke.log
      tail -n 3 gpurun_out/${TAG}_smoke.log ;;
    bench)
      START=$(date +%s)
      timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
      echo "rc=$? wall=$(( $(date +%s) - START ))s" >> gpurun_out/${TAG}_bench.err; tail -n 2 gpurun_out/${TAG}_bench.err
      python tools/bench_digest.py gpurun_out/${TAG}_bench.json ;;
    benchq)     # the headline loop only (no strict / vendor / cpu legs): quick A/B of `value`
      timeout 600 python bench.py --steps 20 --warmup 5 $QUIET > gpurun_out/${TAG}_benchq.json 2> gpurun_out/${TAG}_benchq.err
      python tools/bench_digest.py gpurun_out/${TAG}_benchq.json ;;
    benchab:*)  # in-step A/B of a developer debug bit: benchab:<bits> runs bits / 1073741824 (a no-op bit, same library) alternately, 3 rounds
      bits=${step#benchab:}
      for r in 1 2 3; do for b in $bits 1073741824; do
        timeout 300 python bench.py --steps 15 --warmup 4 --dev-debug $b $QUIET > gpurun_out/ab.json 2>/dev/null
        python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('debug bits %d: %.1f samples/s  %.3f ms/step (median %.3f)' % ($b, d['value'], d['ms_per_step'], d['ms_per_step_median']))" | tee -a gpurun_out/${TAG}_benchab_$bits.txt
      done; done ;;
    stats)
      timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --steps 8 --warmup 2 $QUIET > gpurun_out/pf.log 2>&1
      python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/${TAG}_kernel_stats_b1024.txt 2>&1; rm -rf gpurun_out/pf
      head -n 30 gpurun_out/${TAG}_kernel_stats_b1024.txt | cut -c1-200 ;;
    statsx3)
      timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --dtype bf16x3 --batch 1024 --steps 8 --warmup 2 $QUIET > gpurun_out/pf2.log 2>&1
      python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/${TAG}_kernel_stats_bf16x3_b1024.txt 2>&1; rm -rf gpurun_out/pf
      head -n 24 gpurun_out/${TAG}_kernel_stats_bf16x3_b1024.txt | cut -c1-200 ;;
    pmc)        # per-kernel MFMA busy / VALU:MFMA / issue stalls / LDS conflicts of the bf16 step (two PMC passes, kernel trace only)
      PQ="--steps 2 --warmup 1 $QUIET"
      SETA="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
      SETB="GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_INST_LEVEL_VMEM"
      timeout 400 rocprofv3 --kernel-trace --pmc $SETA -d gpurun_out/pa -o p -- python bench.py --batch 1024 $PQ > gpurun_out/pa.log 2>&1
      timeout 400 rocprofv3 --kernel-trace --pmc $SETB -d gpurun_out/pb -o p -- python bench.py --batch 1024 $PQ > gpurun_out/pb.log 2>&1
      A=$(find gpurun_out/pa -name "*_results.db" | head -1); B=$(find gpurun_out/pb -name "*_results.db" | head -1)
      echo "# python bench.py --dtype bf16 --batch 1024 (3 steps), rocprofv3 --kernel-trace --pmc, two passes; tools/pmc_step_summary.py" > gpurun_out/${TAG}_pmc_step_bf16.txt
      python tools/pmc_step_summary.py $A $B >> gpurun_out/${TAG}_pmc_step_bf16.txt 2>&1
      rm -rf gpurun_out/pa gpurun_out/pb
      head -n 16 gpurun_out/${TAG}_pmc_step_bf16.txt | cut -c1-200 ;;
    pmcb:*)     # the same two PMC passes at another per-GPU batch: pmcb:<B>  (round 6: B = 8, the small-batch kernels)
      PB=${step#pmcb:}
      PQ="--steps 6 --warmup 2 $QUIET"
      SETA="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
      SETB="GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_INST_LEVEL_VMEM"
      timeout 400 rocprofv3 --kernel-trace --pmc $SETA -d gpurun_out/pa -o p -- python bench.py --batch $PB $PQ > gpurun_out/pa.log 2>&1
      timeout 400 rocprofv3 --kernel-trace --pmc $SETB -d gpurun_out/pb -o p -- python bench.py --batch $PB $PQ > gpurun_out/pb.log 2>&1
      A=$(find gpurun_out/pa -name "*_results.db" | head -1); B=$(find gpurun_out/pb -name "*_results.db" | head -1)
      echo "# python bench.py --dtype bf16 --batch $PB (8 steps), rocprofv3 --kernel-trace --pmc, two passes; tools/pmc_step_summary.py" > gpurun_out/${TAG}_pmc_step_bf16_b$PB.txt
      python tools/pmc_step_summary.py $A $B >> gpurun_out/${TAG}_pmc_step_bf16_b$PB.txt 2>&1
      rm -rf gpurun_out/pa gpurun_out/pb
      head -n 22 gpurun_out/${TAG}_pmc_step_bf16_b$PB.txt | cut -c1-200 ;;
    pyt:*)      # a subset of the GPU suite: pyt:<-k expression with '+' for spaces>
      expr=$(echo "${step#pyt:}" | tr '+' ' ')
      timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "$expr" > gpurun_out/${TAG}_pytest_subset.log 2>&1
      echo "rc=$?" >> gpurun_out/${TAG}_pytest_subset.log; tail -n 6 gpurun_out/${TAG}_pytest_subset.log ;;
    gemmab:*)
      arms=$(echo "${step#gemmab:}" | tr ',' ' ')
      VB_DEV=1 timeout 400 python tools/gemm_ab.py 1024 $arms > gpurun_out/${TAG}_gemm_ab.txt 2>&1; cut -c1-260 gpurun_out/${TAG}_gemm_ab.txt ;;
    attn)
      timeout 300 python tools/attn_bench.py 1024 164 > gpurun_out/${TAG}_attn_bench.txt 2>&1; tail -n 12 gpurun_out/${TAG}_attn_bench.txt ;;
    nlvr2real)
      timeout 600 python bench.py --workload nlvr2-real --steps 10 --warmup 3 --no-cpu-baseline --strict-dtype none --no-vendor-leg --pmc-traffic off > gpurun_out/${TAG}_bench_nlvr2_real.json 2> gpurun_out/${TAG}_bench_nlvr2_real.err
      python tools/bench_digest.py gpurun_out/${TAG}_bench_nlvr2_real.json
      timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --workload nlvr2-real --steps 4 --warmup 2 $QUIET > gpurun_out/pf3.log 2>&1
      python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/${TAG}_kernel_stats_nlvr2_real.txt 2>&1; rm -rf gpurun_out/pf
      head -n 16 gpurun_out/${TAG}_kernel_stats_nlvr2_real.txt | cut -c1-200 ;;
    workloads)  # the other BASELINE configs' quiet bench lines: VQA (configs[3]) and NLVR2 (configs[4])
      for wlk in vqa nlvr2; do
        timeout 600 python bench.py --workload $wlk --steps 20 --warmup 5 $QUIET > gpurun_out/${TAG}_bench_$wlk.json 2> gpurun_out/${TAG}_bench_$wlk.err
        python tools/bench_digest.py gpurun_out/${TAG}_bench_$wlk.json | head -n 2
      done ;;
    sweep)
      timeout 900 python tools/batch_sweep.py > gpurun_out/${TAG}_batch_sweep.txt 2> gpurun_out/${TAG}_batch_sweep.err; cat gpurun_out/${TAG}_batch_sweep.txt ;;
    py:*)
      cmd=$(echo "${step#py:}" | tr ':' ' ')
      name=$(echo "${step#py:}" | tr -c 'A-Za-z0-9_\n' '_' | cut -c1-60)
      timeout 600 python $cmd > gpurun_out/${TAG}_${name}.txt 2>&1; echo "rc=$?"; tail -n 40 gpurun_out/${TAG}_${name}.txt | cut -c1-260 ;;
    *) echo "unknown step $step" ;;
  esac
done
