#!/usr/bin/env python
"""GEMM micro-benchmark on the GPU box: sweeps the pipelined-kernel variants over the BERT-base shapes.
usage: python tools/gemm_bench.py [batch]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visualbert_amd import _lib, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
M = B * 164
dev = torch.device("cuda", 0)
import _knobs
L = _knobs.L
shapes = [("qkv fwd", M, 2304, 768), ("attn-out fwd", M, 768, 768), ("ffn-in fwd(gelu)", M, 3072, 768), ("ffn-in fwd(savegrad)", M, 3072, 768),
          ("ffn-out fwd", M, 768, 3072), ("qkv dgrad", M, 768, 2304), ("decoder fwd f32", M, 30522, 768)]


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


g = torch.Generator().manual_seed(0)
print("M=%d" % M)
for name, m, n, k in shapes:
    a = (torch.randn(m, k, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    w = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    bias = torch.randn(n, generator=g).to(dev)
    f32out = "f32" in name
    out = ops.alloc2d(m, n, torch.float32 if f32out else torch.bfloat16, dev)
    pre = torch.empty(m, n, dtype=torch.bfloat16, device=dev) if ("gelu" in name or "savegrad" in name) else None
    actc = 4 if "savegrad" in name else (1 if "gelu" in name else 0)
    row = []
    outs = {}
    for v in (42, 80, 81):
        assert _knobs.variant(v) == 0
        def fn():
            ops.gemm(a, w, m, n, k, out=out, bias=bias, act=actc, aux_out=pre)
        ms = bench(fn)
        row.append("v%d %6.1f us %6.0f TF" % (v, ms * 1e3, 2.0 * m * n * k / ms / 1e9))
        outs[v] = out[:, :n].float().clone() if n < 8192 else out[::7, :n:5].float().clone()
    dif = max((outs[v] - outs[42]).abs().max().item() for v in outs)     # all kernels must agree (async-copy race screen)
    print("%-18s N=%5d K=%5d | %s | maxdiff %.2e" % (name, n, k, " | ".join(row), dif))
_knobs.variant(1); L.vb_gemm_set_debug(0)
# wgrad (both K-strided, split-K)
for name, n_out, k_in in [("wgrad qkv", 2304, 768), ("wgrad attn-out", 768, 768), ("wgrad ffn-in", 3072, 768), ("wgrad ffn-out", 768, 3072)]:
    dy = (torch.randn(M, n_out, generator=g) * 0.1).to(torch.bfloat16).to(dev)
    x = (torch.randn(M, k_in, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    dw = torch.zeros(n_out, k_in, device=dev)
    ms = bench(lambda: ops.linear_wgrad(dy, x, dw))
    print("%-18s out=%5d in=%5d | %6.1f us %6.0f TF" % (name, n_out, k_in, ms * 1e3, 2.0 * M * n_out * k_in / ms / 1e9))
