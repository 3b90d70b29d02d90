#!/usr/bin/env python
"""in-kernel timeline of the pipelined GEMM (waves 0 and 4 of workgroup 0): python tools/gemm_trace.py N K"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visualbert_amd import _lib, ops
n, k = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda", 0)
import _knobs
L = _knobs.L
M = 64 * 164
g = torch.Generator().manual_seed(0)
a = (torch.randn(M, k, generator=g) * 0.5).to(torch.bfloat16).to(dev)
w = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16).to(dev)
out = torch.empty(M, n, dtype=torch.bfloat16, device=dev)
tr = torch.zeros(2 * 64 * 8, dtype=torch.int64, device=dev)
dbg = int(sys.argv[3]) if len(sys.argv) > 3 else 64
_knobs.variant(42); L.vb_gemm_set_trace(_lib.ptr(tr)); L.vb_gemm_set_debug(dbg)
for _ in range(3):
    ops.gemm(a, w, M, n, k, out=out)
torch.cuda.synchronize()
L.vb_gemm_set_debug(0)
t = tr.view(2, 64, 8).cpu()
nk = k // 64
names = ["wait->barrier", "barrier->issued", "issued->frags0", "frags0->mfma0 issued", "mfma0->frags1", "frags1->mfma1 issued", "mfma1->next landed"]
for wv in (0, 1):
    x = t[wv, :nk].double()
    d = [x[:, 1] - x[:, 0], x[:, 2] - x[:, 1], x[:, 3] - x[:, 2], x[:, 4] - x[:, 3], x[:, 5] - x[:, 4], x[:, 6] - x[:, 5]]
    nxt = x[1:, 0] - x[:-1, 6]
    per = x[1:, 0] - x[:-1, 0]
    print("dbg %d " % dbg + "wave %d: iteration period %.0f cycles (median); segments (median cycles): %s | mfma1->next landed %.0f" % (
        wv * 4, per.median(), ", ".join("%s %.0f" % (nm, v.median()) for nm, v in zip(names, d)), nxt.median()))
