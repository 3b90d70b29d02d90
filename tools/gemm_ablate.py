#!/usr/bin/env python
"""which of {copies, fragment reads, MFMAs} bounds a GEMM kernel: timing-only builds with parts removed (developer library,
vb_gemm_set_debug bits; results are wrong by construction).  The four-wave kernel (nt_kernel 100) has every combination
(1 no copies, 2 no reads, 3 MFMAs only, 4 no MFMAs, 5 reads only, 6 copies only, 7 copies only without the per-K-tile
drain); the older kernels the single bits 1 / 2 / 4.  profiles/r02_gemm_big_tile_notes.txt is written from this.
    VB_DEV=1 python tools/gemm_ablate.py [kernel ids ...]      default: 100      (VB_BATCH=512)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visualbert_amd import _lib, ops
dev = torch.device("cuda", 0)
import _knobs
L = _knobs.L
M = int(os.environ.get('VB_BATCH', '512')) * 164
VARIANTS = [int(x) for x in sys.argv[1:]] or [100]
g = torch.Generator().manual_seed(0)
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for name, n, k in [("ffn-out", 768, 3072), ("qkv", 2304, 768)]:
    a = (torch.randn(M, k, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    w = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    out = torch.empty(M, n, dtype=torch.bfloat16, device=dev)
    for v in VARIANTS:
        _knobs.variant(v)
        row = []
        for dbg, label in [(0, "full"), (1, "no copies"), (2, "no reads"), (3, "MFMAs only"), (4, "no MFMAs"), (5, "reads only"), (6, "copies only"), (7, "copies only, no drain")] if v == 100 else [(0, "full"), (1, "no tile loads"), (2, "no fragment reads"), (4, "no MFMAs")]:
            L.vb_gemm_set_debug(dbg)
            ms = bench(lambda: ops.gemm(a, w, M, n, k, out=out))
            row.append("%s %.1fus" % (label, ms * 1e3))
        L.vb_gemm_set_debug(0)
        ref = a.float() @ w.float().t()
        ops.gemm(a, w, M, n, k, out=out)
        err = (out.float() - ref).abs().max().item() / ref.abs().max().item()
        row.append("relerr %.1e" % err)
        print("%-8s N=%d K=%d v%d | %s" % (name, n, k, v, " | ".join(row)))
