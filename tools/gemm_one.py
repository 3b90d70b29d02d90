#!/usr/bin/env python
"""one GEMM shape, one variant, a few launches (for PMC runs): python tools/gemm_one.py N K variant [debug]
env: M_ROWS (default 128*164), WGS, EPI = plain | gelu (bias + GELU + saved GELU') | mulaux (x saved GELU' + column sums) | add"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visualbert_amd import _lib, ops
n, k, v = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dbg = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = torch.device("cuda", 0)
import _knobs
L = _knobs.L
M = int(os.environ.get("M_ROWS", 128 * 164))
epi = os.environ.get("EPI", "plain")
g = torch.Generator().manual_seed(0)
a = (torch.randn(M, k, generator=g) * 0.5).to(torch.bfloat16).to(dev)
w = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16).to(dev)
out = torch.empty(M, n, dtype=torch.bfloat16, device=dev)
kw = dict(out=out)
if epi == "gelu":
    kw.update(bias=torch.randn(n, generator=g).to(dev), act=_lib.VB_ACT_GELU_SAVE_GRAD, aux_out=torch.empty(M, n, dtype=torch.bfloat16, device=dev))
elif epi == "mulaux":
    kw.update(act=_lib.VB_ACT_MUL_AUX, aux_in=torch.randn(M, n, device=dev).to(torch.bfloat16), colsum_out=torch.zeros(n, device=dev))
elif epi == "add":
    kw.update(addend=torch.randn(M, n, device=dev).to(torch.bfloat16))
elif epi == "bias":
    kw.update(bias=torch.randn(n, generator=g).to(dev))
_knobs.variant(v); L.vb_gemm_set_debug(dbg); _knobs.wgs(int(os.environ.get("WGS", 0)))
for _ in range(5):
    ops.gemm(a, w, M, n, k, **kw)
torch.cuda.synchronize()
