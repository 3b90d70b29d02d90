#!/usr/bin/env python
"""Why the one-pass attention backward is slower inside the training step than alone.
HIP events around the attention backward only, (a) back to back with itself, (b) behind the kernels that precede it in
vb_bert_layer_bwd (LayerNorm backward, attention-out dgrad GEMM) and in front of the QKV dgrad + grouped wgrad that follow
it, (c) like (b) but with the whole of a layer's backward GEMM work in front (the power / clock state of the real step).
usage: python tools/attn_instep.py [batch]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visualbert_amd import _lib, ops

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
S, nh, H, I = 164, 12, 768, 3072
M = B * S
g = torch.Generator().manual_seed(0)
bf = torch.bfloat16
qkv = (0.5 * torch.randn(M, 3 * H, generator=g)).to(bf).to(dev)
mask = torch.zeros(B, S, device=dev)
dao = torch.randn(M, H, generator=g).to(bf).to(dev)
wo_t = (0.05 * torch.randn(H, H, generator=g)).to(bf).to(dev)
wqkv_t = (0.05 * torch.randn(H, 3 * H, generator=g)).to(bf).to(dev)
wi_t = (0.05 * torch.randn(H, I, generator=g)).to(bf).to(dev)
wo2_t = (0.05 * torch.randn(I, H, generator=g)).to(bf).to(dev)
dfo = torch.randn(M, H, generator=g).to(bf).to(dev)
pre = torch.randn(M, I, generator=g).to(bf).to(dev)
dz = torch.randn(M, H, generator=g).to(bf).to(dev)
p = 0.1
ctx, lse, bits = ops.attn_fwd(qkv, mask, B, S, nh, p, 5, 3)
db = torch.zeros(3 * H, device=dev)
dctx = torch.empty(M, H, dtype=bf, device=dev)
dpre = torch.empty(M, I, dtype=bf, device=dev)
da = torch.empty(M, H, dtype=bf, device=dev)
dh = torch.empty(M, H, dtype=bf, device=dev)
colsum = torch.zeros(I, device=dev)


def attn():
    return ops.attn_bwd(qkv, mask, dctx, lse, bits, B, S, nh, p, 5, 3, ctx_fwd=ctx, dqkv_bias=db)


def measure(before, after, iters=12):
    evs = []
    for it in range(iters + 2):
        before()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dqkv = attn()
        e1.record()
        after(dqkv)
        if it >= 2:
            evs.append((e0, e1))
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
    return ts[len(ts) // 2], ts[0], ts[-1]


def nothing(*a):
    pass


def dgrad_ao():
    ops.gemm(dao, wo_t, M, H, H, out=dctx)


def ffn_bwd_gemms():
    ops.gemm(dfo, wo2_t, M, I, H, out=dpre, act=_lib.VB_ACT_MUL_AUX, aux_in=pre, colsum_out=colsum)
    ops.gemm(dpre, wi_t, M, H, I, out=da, addend=dz)
    dgrad_ao()


def qkv_dgrad(dqkv):
    ops.gemm(dqkv, wqkv_t, M, H, 3 * H, out=dh, addend=dz)


dgrad_ao()
print("B=%d (M=%d): one-pass attention backward + bias gradient, HIP events around the kernel (+ its 6 us reduction)" % (B, M))
print("  alone, back to back             : median %.1f us (min %.1f, max %.1f)" % measure(nothing, nothing))
print("  behind the attention-out dgrad  : median %.1f us (min %.1f, max %.1f)" % measure(dgrad_ao, nothing))
print("  dgrad before, QKV dgrad after   : median %.1f us (min %.1f, max %.1f)" % measure(dgrad_ao, qkv_dgrad))
print("  FFN dgrads + dgrad before, after: median %.1f us (min %.1f, max %.1f)" % measure(ffn_bwd_gemms, qkv_dgrad))
# the same with an idle gap in front of the attention kernel (the chip cools / clocks up): host sleep between before() and attn
import time


def ffn_then_sleep():
    ffn_bwd_gemms()
    torch.cuda.synchronize()
    time.sleep(0.02)


print("  FFN dgrads, 20 ms idle, attention: median %.1f us (min %.1f, max %.1f)" % measure(ffn_then_sleep, nothing))
