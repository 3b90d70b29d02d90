#!/usr/bin/env python
"""fixed (prologue + epilogue) cost of the GEMM kernels: time vs K for the BERT-base N's"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visualbert_amd import _lib, ops
dev = torch.device("cuda", 0)
import _knobs
L = _knobs.L
M = int(sys.argv[1]) * 164 if len(sys.argv) > 1 else 64 * 164
g = torch.Generator().manual_seed(0)
def bench(fn, iters=30):
    """kernel time from the in-library HIP events around each launch (immune to host launch overhead)"""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ops.gemm_profile_start()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    summ = ops.gemm_profile_stop()
    d = list(summ.values())[0]
    return d["ms"] / d["launches"] * 1e3
for v, dbg in ((80, 0), (81, 0)):
    _knobs.variant(v); L.vb_gemm_set_debug(dbg)
    for n in (768, 2304, 3072):
        row = ["dbg=%d" % dbg]
        for k in (64, 128, 256, 768, 1536, 3072):
            a = (torch.randn(M, k, generator=g) * 0.5).to(torch.bfloat16).to(dev)
            w = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16).to(dev)
            bias = torch.randn(n, generator=g).to(dev)
            out = torch.empty(M, n, dtype=torch.bfloat16, device=dev)
            row.append("K=%d %.1fus" % (k, bench(lambda: ops.gemm(a, w, M, n, k, out=out, bias=bias))))
        print("v%d N=%4d | %s" % (v, n, " | ".join(row)))
