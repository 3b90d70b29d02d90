#!/usr/bin/env python
"""what the epilogue's global stores cost on the step's NT shapes (debug bit 128 = same kernel, stores predicated off):
python tools/gemm_store_cost.py [batch]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visualbert_amd import _lib, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
import _knobs
L = _knobs.L
M = B * 164
g = torch.Generator().manual_seed(0)
shapes = [("decoder fwd f32", 30522, 768, True, 0), ("ffn-in fwd gelu", 3072, 768, False, _lib.VB_ACT_GELU_SAVE_GRAD),
          ("qkv fwd", 2304, 768, False, 0), ("attn-out fwd", 768, 768, False, 0), ("ffn-out fwd", 768, 3072, False, 0)]
for name, n, k, f32, act in shapes:
    a = (torch.randn(M, k, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    ld = (n + 7) // 8 * 8
    w = torch.zeros(ld, k, dtype=torch.bfloat16, device=dev)
    w[:n] = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    bias = torch.randn(n, generator=g).to(dev)
    out = torch.empty(M, ld, dtype=torch.float32 if f32 else torch.bfloat16, device=dev)[:, :n]
    aux = torch.empty(M, ld, dtype=torch.bfloat16, device=dev)[:, :n] if act else None
    res = {}
    for rep in range(3):
        for dbg in (0, 128):
            L.vb_gemm_set_debug(dbg)
            for _ in range(2):
                ops.gemm(a, w[:n], M, n, k, out=out, bias=bias, act=act, aux_out=aux)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.gemm(a, w[:n], M, n, k, out=out, bias=bias, act=act, aux_out=aux)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(dbg, []).append(e0.elapsed_time(e1) / 5 * 1e3)
    L.vb_gemm_set_debug(0)
    fl = 2.0 * M * n * k
    print("%-18s %7.1f us (%6.1f TF)   without the stores %7.1f us (%6.1f TF)" % (
        name, min(res[0]), fl / min(res[0]) / 1e6, min(res[128]), fl / min(res[128]) / 1e6))
    del a, w, out, aux
