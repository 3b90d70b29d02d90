import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualbert_amd import _lib, ops
emu = os.environ.get("VB_EMU") == "1"
if emu:
    _lib.set_library(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'hipemu', 'libvisualbert_emu.so'), 'cpu'); dev = torch.device('cpu')
else:
    dev = torch.device('cuda', 0)
nh = 12; H = nh * 64
worst = 0.0
for (B, S, p) in ((3, 164, 0.1), (2, 100, 0.0), (5, 37, 0.1), (2, 176, 0.1), (1, 65, 0.0)) if emu else ((64, 164, 0.1), (33, 100, 0.0), (70, 37, 0.1), (300, 176, 0.1), (1024, 164, 0.1)):
    g = torch.Generator().manual_seed(B * 1000 + S)
    qkv = (0.5 * torch.randn(B * S, 3 * H, generator=g)).to(torch.bfloat16).to(dev)
    mask = torch.zeros(B, S, device=dev); mask[0, S - 3:] = -10000.0
    dctx = torch.randn(B * S, H, generator=g).to(torch.bfloat16).to(dev)
    ctx, lse, bits = ops.attn_fwd(qkv, mask, B, S, nh, p, 5, 3)
    outs = []
    for tp in (0, 2):
        db = torch.zeros(3 * H, device=dev)
        with _lib.stream_opts(attn_two_pass=tp):
            d = ops.attn_bwd(qkv, mask, dctx, lse, bits, B, S, nh, p, 5, 3, ctx_fwd=ctx, dqkv_bias=db)
        outs.append((d.float().clone(), db.clone()))
    e1 = (outs[0][0] - outs[1][0]).abs().max().item(); e2 = (outs[0][1] - outs[1][1]).abs().max().item()
    print("B=%d S=%d p=%.1f: max|d dqkv| %.3e (max %.3e)  max|d bias| %.3e (max %.3e)" % (B, S, p, e1, outs[0][0].abs().max().item(), e2, outs[0][1].abs().max().item()))
    assert torch.equal(outs[0][0], outs[1][0]), "dqkv differs"
    assert e2 <= 1e-5 * max(1.0, outs[0][1].abs().max().item())
print("persistent == per-pair kernel")
