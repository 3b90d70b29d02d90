import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visualbert_amd import _lib, ops
if os.environ.get("VB_LIB_PATH"):                       # A/B of two product builds: VB_LIB_PATH=tools/libvisualbert_hip_ab_<arm>.so
    _lib.set_library(os.path.abspath(os.environ["VB_LIB_PATH"]))
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S, nh = (int(sys.argv[2]) if len(sys.argv) > 2 else 164), 12
H = nh * 64
g = torch.Generator().manual_seed(0)
qkv = (0.5 * torch.randn(B * S, 3 * H, generator=g)).to(torch.bfloat16).to(dev)
mask = torch.zeros(B, S, device=dev)
dctx = torch.randn(B * S, H, generator=g).to(torch.bfloat16).to(dev)
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
flops = 4.0 * S * S * 64 * nh * B
for p in (0.0, 0.1):
    ctx, lse, bits = ops.attn_fwd(qkv, mask, B, S, nh, p, 5, 3)
    tf = bench(lambda: ops.attn_fwd(qkv, mask, B, S, nh, p, 5, 3))
    tf1 = bench(lambda: ops.attn_fwd(qkv, mask, B, S, nh, p, 5, 3))
    print("B=%d p=%.1f: fwd %.1f / %.1f us" % (B, p, tf, tf1))
    tb = bench(lambda: ops.attn_bwd(qkv, mask, dctx, lse, bits, B, S, nh, p, 5, 3))
    t1 = bench(lambda: ops.attn_bwd(qkv, mask, dctx, lse, bits, B, S, nh, p, 5, 3, ctx_fwd=ctx))
    tb2 = bench(lambda: ops.attn_bwd(qkv, mask, dctx, lse, bits, B, S, nh, p, 5, 3))
    t12 = bench(lambda: ops.attn_bwd(qkv, mask, dctx, lse, bits, B, S, nh, p, 5, 3, ctx_fwd=ctx))
    db = torch.zeros(3 * H, device=dev)
    t1b = bench(lambda: ops.attn_bwd(qkv, mask, dctx, lse, bits, B, S, nh, p, 5, 3, ctx_fwd=ctx, dqkv_bias=db))
    print("B=%d p=%.1f: one-pass backward + q|k|v bias gradient %.1f us" % (B, p, t1b))
    print("B=%d p=%.1f: fwd %.1f us (%.0f TF)  bwd two-pass %.1f / %.1f us (%.0f TF)  bwd one-pass %.1f / %.1f us" % (
        B, p, tf, flops / tf / 1e6, tb, tb2, 2.5 * flops / tb / 1e6, t1, t12))
