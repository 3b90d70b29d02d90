#!/bin/bash
# The one-off GPU sessions of round 6, one case each (they were tools/gpu_r06_s<N>.sh): the exact commands behind profiles/r06_small_batch_ab.txt,
# r06_attn_pitch_ab.txt, r06_dropres_epilogue_ab.txt, r06_gelu_polynomial_ab.txt and the r06_s<N>_* kernel statistics.  A RECORD, not a harness: several
# cases name A/B libraries (tools/libvisualbert_hip_ab_<arm>.so, built with tools/build_variant.sh and the flags the case states) or bench.py flags of
# experiments that were removed again after they lost.
#     gpurun --timeout 1500 -- 'bash tools/gpu_r06_sessions.sh s13'
case "$1" in
s1)
# round 6, session 1: kernel tests of the changed kernels, small-batch A/B of two product builds, kernel trace at B = 8
export TMPDIR=/tmp
mkdir -p gpurun_out
START=$(date +%s)
timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/r06_s1_pytest_kernels.log 2>&1
echo "rc=$? wall=$(( $(date +%s) - START ))s" >> gpurun_out/r06_s1_pytest_kernels.log; tail -n 5 gpurun_out/r06_s1_pytest_kernels.log
bash tools/gpu_small_batch_ab.sh tools/libvisualbert_hip_ab_base.so 8 16 32 64 2>&1 | tee gpurun_out/r06_s1_small_batch_ab.txt
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
for B in 8 32; do
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --batch $B --steps 20 --warmup 5 $QUIET > gpurun_out/pf_b$B.log 2>&1
python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/r06_s1_kernel_stats_b$B.txt 2>&1; rm -rf gpurun_out/pf
head -n 12 gpurun_out/r06_s1_kernel_stats_b$B.txt | cut -c1-200
done
echo "total wall=$(( $(date +%s) - START ))s"
;;
s2)
# round 6, session 2: first split-K (plain slab stores + agent-scope release / acquire fences) + the decoder weight gradient on the small-token kernel: kernel tests, small-batch A/B, kernel traces at B = 8 / 16
export TMPDIR=/tmp
mkdir -p gpurun_out
START=$(date +%s)
timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/r06_s2_pytest_kernels.log 2>&1
echo "rc=$? wall=$(( $(date +%s) - START ))s" >> gpurun_out/r06_s2_pytest_kernels.log; tail -n 5 gpurun_out/r06_s2_pytest_kernels.log
bash tools/gpu_small_batch_ab.sh tools/libvisualbert_hip_ab_base.so 8 16 32 2>&1 | tee gpurun_out/r06_s2_small_batch_ab.txt
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
for B in 8 16; do
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --batch $B --steps 20 --warmup 5 $QUIET > gpurun_out/pf_b$B.log 2>&1
python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/r06_s2_kernel_stats_b$B.txt 2>&1; rm -rf gpurun_out/pf
head -n 12 gpurun_out/r06_s2_kernel_stats_b$B.txt | cut -c1-200
done
echo "total wall=$(( $(date +%s) - START ))s"
;;
s3)
# round 6, session 3: split-K with write-through hand-off; three product builds (base = round 5 kernels, nosk = + small-token wgrad, in-tree = + split-K)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider -k "split_k or wgrad or gemm" > gpurun_out/r06_s3_pytest_kernels.log 2>&1
echo "rc=$?" >> gpurun_out/r06_s3_pytest_kernels.log; tail -n 4 gpurun_out/r06_s3_pytest_kernels.log
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
for B in 8 16; do for r in 1 2; do for lib in tools/libvisualbert_hip_ab_base.so tools/libvisualbert_hip_ab_nosk.so visualbert_amd/libvisualbert_hip.so; do
  timeout 300 python bench.py --batch $B --steps 30 --warmup 8 --lib-path $lib $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('B=%4d $lib: %.1f samples/s  %.3f ms/step (median %.3f)' % ($B, d['value'], d['ms_per_step'], d['ms_per_step_median']))" || tail -3 gpurun_out/ab.err
done; done; done 2>&1 | tee gpurun_out/r06_s3_small_batch_ab.txt
for B in 8; do
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --batch $B --steps 20 --warmup 5 $QUIET > gpurun_out/pf_b$B.log 2>&1
python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/r06_s3_kernel_stats_b$B.txt 2>&1; rm -rf gpurun_out/pf
head -n 12 gpurun_out/r06_s3_kernel_stats_b$B.txt | cut -c1-200
done
;;
s5)
# round 6, session 5: host-side call plans -- host profile at B = 8, then nosk / in-tree product builds at B = 8 / 16 / 32
export TMPDIR=/tmp
mkdir -p gpurun_out
python tools/host_profile.py --batch 8 > gpurun_out/r06_host_profile_b8_plans.txt 2>&1; head -30 gpurun_out/r06_host_profile_b8_plans.txt | cut -c1-150
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
for B in 8 16 32; do for r in 1 2; do for lib in tools/libvisualbert_hip_ab_nosk.so visualbert_amd/libvisualbert_hip.so; do
  timeout 300 python bench.py --batch $B --steps 30 --warmup 8 --lib-path $lib $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('B=%4d $lib: %.1f samples/s  %.3f ms/step (median %.3f)' % ($B, d['value'], d['ms_per_step'], d['ms_per_step_median']))" || tail -3 gpurun_out/ab.err
done; done; done 2>&1 | tee gpurun_out/r06_s5_small_batch_ab.txt
timeout 600 python -m pytest tests/test_model_parity.py tests/test_parity_at_scale.py -m gpu -q -x --tb=short -p no:cacheprovider > gpurun_out/r06_s5_pytest_parity.log 2>&1
echo "rc=$?" >> gpurun_out/r06_s5_pytest_parity.log; tail -n 4 gpurun_out/r06_s5_pytest_parity.log
;;
s6)
# round 6, session 6: where the small-token weight-gradient kernel stops winning (48 / 96 / 192 K tiles) at B = 24 ... 128
export TMPDIR=/tmp
mkdir -p gpurun_out
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
for B in 24 32 48 64 128; do for r in 1 2; do for lib in visualbert_amd/libvisualbert_hip.so tools/libvisualbert_hip_ab_kt96.so tools/libvisualbert_hip_ab_kt192.so; do
  timeout 300 python bench.py --batch $B --steps 30 --warmup 8 --lib-path $lib $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('B=%4d $lib: %.1f samples/s  %.3f ms/step (median %.3f)' % ($B, d['value'], d['ms_per_step'], d['ms_per_step_median']))" || tail -3 gpurun_out/ab.err
done; done; done 2>&1 | tee gpurun_out/r06_s6_wgrad_crossover.txt
;;
s7)
# round 6, session 7: dropout + residual in the producing GEMM's epilogue -- parity at the bench shape, then product builds A/B at B = 1024
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_bench_shape.py tests/test_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider -k "layer or split_k or wgrad" > gpurun_out/r06_s7_pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r06_s7_pytest.log; tail -n 6 gpurun_out/r06_s7_pytest.log
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
for r in 1 2 3; do for lib in tools/libvisualbert_hip_ab_nofuse.so visualbert_amd/libvisualbert_hip.so tools/libvisualbert_hip_ab_fuse90.so; do
  timeout 300 python bench.py --steps 15 --warmup 4 --lib-path $lib $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('$lib: %.1f samples/s  %.3f ms/step (median %.3f)' % (d['value'], d['ms_per_step'], d['ms_per_step_median']))" || tail -3 gpurun_out/ab.err
done; done 2>&1 | tee gpurun_out/r06_s7_dropres_ab.txt
for lib in tools/libvisualbert_hip_ab_nofuse.so visualbert_amd/libvisualbert_hip.so; do
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --steps 8 --warmup 2 --lib-path $lib $QUIET > gpurun_out/pf.log 2>&1
  python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/r06_s7_stats_$(basename $lib .so).txt 2>&1; rm -rf gpurun_out/pf
  echo "$lib: $(head -1 gpurun_out/r06_s7_stats_$(basename $lib .so).txt)"; grep -E "gemm_nt|ln_fwd|ln_bwd" gpurun_out/r06_s7_stats_$(basename $lib .so).txt | cut -c1-175 | head -14
done
;;
s8)
# round 6, session 8: LDS pitches of the attention kernels' transposed tiles (8 -> 16 bytes of row padding: conflict-free 8-byte fragment reads)
#   attn_old = rounds 2-5 (TPAD 8, dS / K^T pitch 392) | attn_b = dS / K^T pitch 400 only | in-tree = every transposed bf16 tile padded by 16
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider -k "attention" > gpurun_out/r06_s8_pytest_attn.log 2>&1
echo "rc=$?" >> gpurun_out/r06_s8_pytest_attn.log; tail -n 4 gpurun_out/r06_s8_pytest_attn.log
for r in 1 2; do for lib in tools/libvisualbert_hip_ab_attn_old.so tools/libvisualbert_hip_ab_attn_b.so visualbert_amd/libvisualbert_hip.so; do
  echo "== $lib"; VB_LIB_PATH=$lib timeout 300 python tools/attn_bench.py 1024 164 2>&1 | grep -E "p=0.1" | grep -E "one-pass backward|fwd .* us \("
done; done 2>&1 | tee gpurun_out/r06_s8_attn_bench.txt
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
for r in 1 2 3; do for lib in tools/libvisualbert_hip_ab_attn_old.so tools/libvisualbert_hip_ab_attn_b.so visualbert_amd/libvisualbert_hip.so; do
  timeout 300 python bench.py --steps 15 --warmup 4 --lib-path $lib $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('$lib: %.1f samples/s  %.3f ms/step (median %.3f)' % (d['value'], d['ms_per_step'], d['ms_per_step_median']))" || tail -3 gpurun_out/ab.err
done; done 2>&1 | tee gpurun_out/r06_s8_attn_step_ab.txt
for lib in tools/libvisualbert_hip_ab_attn_old.so visualbert_amd/libvisualbert_hip.so; do
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --steps 8 --warmup 2 --lib-path $lib $QUIET > gpurun_out/pf.log 2>&1
  python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/r06_s8_stats_$(basename $lib .so).txt 2>&1; rm -rf gpurun_out/pf
  echo "$lib: $(head -1 gpurun_out/r06_s8_stats_$(basename $lib .so).txt)"; grep -E "attn_" gpurun_out/r06_s8_stats_$(basename $lib .so).txt | cut -c1-175 | head -5
done
;;
s9)
# round 6, session 9: kernel traces of the mid-size batches (B = 64, 128) on the current tree
export TMPDIR=/tmp
mkdir -p gpurun_out
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
for B in 64 128; do
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --batch $B --steps 20 --warmup 5 $QUIET > gpurun_out/pf_b$B.log 2>&1
python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/r06_s9_kernel_stats_b$B.txt 2>&1; rm -rf gpurun_out/pf
head -n 26 gpurun_out/r06_s9_kernel_stats_b$B.txt | cut -c1-200
done
;;
s10)
# round 6, session 10: small-token weight gradients with 32-token K tiles on a four-stage ring (two workgroups per CU) against the two-stage 64-token ring
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider -k "wgrad or split_k" > gpurun_out/r06_s10_pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r06_s10_pytest.log; tail -n 4 gpurun_out/r06_s10_pytest.log
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
for B in 8 16 24 32; do for r in 1 2; do for lib in tools/libvisualbert_hip_ab_kb64.so visualbert_amd/libvisualbert_hip.so; do
  timeout 300 python bench.py --batch $B --steps 30 --warmup 8 --lib-path $lib $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('B=%4d $lib: %.1f samples/s  %.3f ms/step (median %.3f)' % ($B, d['value'], d['ms_per_step'], d['ms_per_step_median']))" || tail -3 gpurun_out/ab.err
done; done; done 2>&1 | tee gpurun_out/r06_s10_kb32_ab.txt
for B in 8; do
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --batch $B --steps 20 --warmup 5 $QUIET > gpurun_out/pf_b$B.log 2>&1
python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/r06_s10_kernel_stats_b$B.txt 2>&1; rm -rf gpurun_out/pf
head -n 14 gpurun_out/r06_s10_kernel_stats_b$B.txt | cut -c1-200
done
;;
s12)
# round 6, session 12: split-K geometry -- 128x128 tiles with more slices (p128), also the K = 768 GEMMs (sk12: slices of >= 6 K tiles), both
export TMPDIR=/tmp
mkdir -p gpurun_out
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
for B in 8 16; do for r in 1 2; do for lib in visualbert_amd/libvisualbert_hip.so tools/libvisualbert_hip_ab_p128.so tools/libvisualbert_hip_ab_sk12.so tools/libvisualbert_hip_ab_p128sk12.so; do
  timeout 300 python bench.py --batch $B --steps 30 --warmup 8 --lib-path $lib $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('B=%4d $lib: %.1f samples/s  %.3f ms/step (median %.3f)' % ($B, d['value'], d['ms_per_step'], d['ms_per_step_median']))" || tail -3 gpurun_out/ab.err
done; done; done 2>&1 | tee gpurun_out/r06_s12_splitk_geometry_ab.txt
for lib in tools/libvisualbert_hip_ab_p128sk12.so; do
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --batch 8 --steps 20 --warmup 5 --lib-path $lib $QUIET > gpurun_out/pf_b8.log 2>&1
python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/r06_s12_kernel_stats_b8_p128sk12.txt 2>&1; rm -rf gpurun_out/pf
head -n 14 gpurun_out/r06_s12_kernel_stats_b8_p128sk12.txt | cut -c1-200
done
;;
s13)
# round 6, session 13: small-token weight gradients on 256x128 tiles with eight waves (in-tree) against the 128x128 kernel (no256)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels.py -m gpu -q -x --tb=short -p no:cacheprovider -k "wgrad" > gpurun_out/r06_s13_pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r06_s13_pytest.log; tail -n 4 gpurun_out/r06_s13_pytest.log
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
for B in 8 16 24 32; do for r in 1 2; do for lib in tools/libvisualbert_hip_ab_no256.so visualbert_amd/libvisualbert_hip.so; do
  timeout 300 python bench.py --batch $B --steps 30 --warmup 8 --lib-path $lib $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('B=%4d $lib: %.1f samples/s  %.3f ms/step (median %.3f)' % ($B, d['value'], d['ms_per_step'], d['ms_per_step_median']))" || tail -3 gpurun_out/ab.err
done; done; done 2>&1 | tee gpurun_out/r06_s13_wgrad256_ab.txt
for B in 8 32; do
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pf -o f -- python bench.py --batch $B --steps 20 --warmup 5 $QUIET > gpurun_out/pf_b$B.log 2>&1
python tools/rocpd_summary.py gpurun_out/pf/f_results.db > gpurun_out/r06_s13_kernel_stats_b$B.txt 2>&1; rm -rf gpurun_out/pf
grep -E "^# kernels|gemm_tn" gpurun_out/r06_s13_kernel_stats_b$B.txt | cut -c1-200
done
;;
s14)
# round 6, session 14: crossover of the 256x128 small-token kernel against the persistent kernel: 96 (in-tree) / 128 / 192 K tiles at B = 40 ... 64
export TMPDIR=/tmp
mkdir -p gpurun_out
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
for B in 8 32 40 48 64; do for r in 1 2; do for lib in visualbert_amd/libvisualbert_hip.so tools/libvisualbert_hip_ab_kt128.so tools/libvisualbert_hip_ab_kt192.so; do
  timeout 300 python bench.py --batch $B --steps 30 --warmup 8 --lib-path $lib $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('B=%4d $lib: %.1f samples/s  %.3f ms/step (median %.3f)' % ($B, d['value'], d['ms_per_step'], d['ms_per_step_median']))" || tail -3 gpurun_out/ab.err
done; done; done 2>&1 | tee gpurun_out/r06_s14_wgrad256_crossover.txt
;;
s15)
# round 6, session 15: the mid-size batches (B = 48 ... 96) with the NT GEMMs pinned to one pipelined kernel each (0 = the rule)
export TMPDIR=/tmp
mkdir -p gpurun_out
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
for B in 48 64 96; do for k in 0 22 42 24 90 81; do
  timeout 300 python bench.py --batch $B --steps 30 --warmup 8 --nt-kernel $k $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('B=%4d nt_kernel %3d: %.1f samples/s  %.3f ms/step (median %.3f)' % ($B, $k, d['value'], d['ms_per_step'], d['ms_per_step_median']))" || tail -3 gpurun_out/ab.err
done; done 2>&1 | tee gpurun_out/r06_s15_mid_batch_nt_kernel.txt
;;
s20)
# round 6, session 20: polynomial GELU pair, per GEMM and per kernel (81 = persistent 256x256, 90 = two-workgroup 256x128): in-tree library
# (VB_GELU_POLY=1) against tools/libvisualbert_hip_ab_nopoly.so, alternating processes; then the step A/B
export TMPDIR=/tmp
mkdir -p gpurun_out
QUIET="--no-cpu-baseline --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg --no-profile"
for r in 1 2; do for arm in poly nopoly; do
  echo "== $arm (round $r)"
  if [ $arm = nopoly ]; then export VB_LIB_PATH=tools/libvisualbert_hip_ab_nopoly.so; else unset VB_LIB_PATH; fi
  VB_NOCHECK=1 timeout 300 python tools/gemm_ab.py 1024 81 90 2>&1 | grep -i "gelu\|per step"
done; done 2>&1 | tee gpurun_out/r06_s20_gelu_poly_gemm_ab.txt
unset VB_LIB_PATH
for r in 1 2 3; do for arm in poly nopoly; do
  lp=""; [ $arm = nopoly ] && lp="--lib-path tools/libvisualbert_hip_ab_nopoly.so"
  timeout 300 python bench.py --steps 15 --warmup 4 $lp $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "
import json;d=json.load(open('gpurun_out/ab.json'))
print('%-7s: %.1f samples/s  %.3f ms/step (median %.3f)' % ('$arm', d['value'], d['ms_per_step'], d['ms_per_step_median']))" || tail -3 gpurun_out/ab.err
done; done 2>&1 | tee gpurun_out/r06_s20_gelu_poly_step_ab.txt
;;
s21)
# round 6, session 21: the layer's three second-stage reductions (LayerNorm x 2, attention bias gradient) as ONE launch at the end of the
# layer's backward, against one launch each (tools/build_variant.sh nodefer "-DVB_DEFER_REDUCE=0"), alternating on one box
export TMPDIR=/tmp
mkdir -p gpurun_out
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off --no-batch-curve --no-sparse-leg"
for r in 1 2 3; do for B in 8 16 32 128; do for arm in defer nodefer; do
  lp=""; [ $arm = nodefer ] && lp="--lib-path tools/libvisualbert_hip_ab_nodefer.so"
  st=40; [ $B -ge 128 ] && st=20
  timeout 300 python bench.py --batch $B --steps $st --warmup 8 $lp $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err
  python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('B=%4d %-8s: %.1f samples/s  %.3f ms/step (median %.3f)' % ($B, '$arm', d['value'], d['ms_per_step'], d['ms_per_step_median']))" || tail -3 gpurun_out/ab.err
done; done; done 2>&1 | tee gpurun_out/r06_s21_defer_reduce_ab.txt
timeout 300 python bench.py --steps 15 --warmup 4 $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err; python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('B=1024 defer: %.1f samples/s %.3f ms' % (d['value'], d['ms_per_step']))" | tee -a gpurun_out/r06_s21_defer_reduce_ab.txt
timeout 300 python bench.py --steps 15 --warmup 4 --lib-path tools/libvisualbert_hip_ab_nodefer.so $QUIET > gpurun_out/ab.json 2>gpurun_out/ab.err; python -c "import json;d=json.load(open('gpurun_out/ab.json'));print('B=1024 nodefer: %.1f samples/s %.3f ms' % (d['value'], d['ms_per_step']))" | tee -a gpurun_out/r06_s21_defer_reduce_ab.txt
timeout 1500 python -m pytest tests/test_kernels.py tests/test_model_parity.py tests/test_parity_at_scale.py tests/test_bench_shape.py -m gpu -q --tb=short -p no:cacheprovider --timeout 900 > gpurun_out/r06_s21_pytest.log 2>&1; tail -n 6 gpurun_out/r06_s21_pytest.log
;;
*) echo "usage: $0 s<N>   (s1 s2 s3 s5 s6 s7 s8 s9 s10 s12 s13 s14 s15 s20 s21)" ;;
esac
