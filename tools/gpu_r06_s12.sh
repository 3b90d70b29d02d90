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
