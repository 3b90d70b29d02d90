# round-4 session 15: HBM traffic of the LayerNorm kernels inside the bf16 step (FETCH_SIZE / WRITE_SIZE, separate PMC passes, kernel trace only)
export TMPDIR=/tmp
mkdir -p gpurun_out
QUIET="--no-cpu-baseline --no-profile --no-h2d --no-parity --strict-dtype none --no-vendor-leg --pmc-traffic off"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pt_$c -o p -- python bench.py --steps 2 --warmup 1 $QUIET > gpurun_out/pt_$c.log 2>&1
  python tools/rocpd_pmc.py gpurun_out/pt_$c/p_results.db > gpurun_out/r15_$c.txt 2>&1
  rm -rf gpurun_out/pt_$c
done
python - <<'PY'
import re
def load(f):
    out={}
    for l in open(f):
        if l.startswith(('#','kernel')): continue
        m=re.match(r"(.{92})\s+(\d+)\s+(\S+)", l)
        if m: out[m.group(1).strip()]=(int(m.group(2)), float(m.group(3)))
    return out
F,W=load("gpurun_out/r15_FETCH_SIZE.txt"),load("gpurun_out/r15_WRITE_SIZE.txt")
print("%-70s %6s %12s %12s" % ("kernel", "calls", "fetched MB", "written MB"), "(per launch; FETCH_SIZE x 1024 x 2, WRITE_SIZE x 1024)")
for k,(n,v) in sorted(F.items(), key=lambda kv:-kv[1][1]):
    if re.search(r"ln_|adam|ce_row|attn|embed", k):
        w=W.get(k,(n,float('nan')))[1]
        print("%-70s %6d %12.1f %12.1f" % (k[:70], n, v*1024*2/n/1e6, w*1024/n/1e6))
PY
