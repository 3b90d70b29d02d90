#!/usr/bin/env python
"""LayerNorm forward / backward timing with and without the dropout sites (what regenerating the mask costs):
python tools/ln_bench.py [batch]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visualbert_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
L = _lib.lib()
M, H = B * 164, 768
g = torch.Generator().manual_seed(0)
dt = torch.bfloat16
x = torch.randn(M, H, generator=g).to(dt).to(dev)
r = torch.randn(M, H, generator=g).to(dt).to(dev)
gamma = torch.ones(H, device=dev); beta = torch.zeros(H, device=dev)
y = torch.empty_like(x); z = torch.empty_like(x); mean = torch.empty(M, device=dev); rstd = torch.empty(M, device=dev)
dy = torch.randn(M, H, generator=g).to(dt).to(dev)
dz = torch.empty_like(x); dx = torch.empty_like(x)
dg = torch.zeros(H, device=dev); db = torch.zeros(H, device=dev); dbias = torch.zeros(H, device=dev)
ws = torch.empty(L.vb_ln_bwd_ws_bytes(M, H) // 4, device=dev)


def timed(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3


for p in (0.0, 0.1):
    fwd = lambda: _lib.check(L.vb_ln_fwd(_lib.VB_BF16, _lib.ptr(x), _lib.ptr(r), _lib.ptr(z), _lib.ptr(y), _lib.ptr(mean),
                                         _lib.ptr(rstd), _lib.ptr(gamma), _lib.ptr(beta), M, H, 1e-12, p, 11, 0.0, 12, 5,
                                         _lib.stream_ptr()), "fwd")
    bwd = lambda: _lib.check(L.vb_ln_bwd(_lib.VB_BF16, _lib.ptr(dy), _lib.ptr(z), _lib.ptr(mean), _lib.ptr(rstd),
                                         _lib.ptr(gamma), _lib.ptr(dz), _lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db),
                                         _lib.ptr(dbias), M, H, p, 11, 0.0, 12, 5, _lib.ptr(ws), _lib.stream_ptr()), "bwd")
    tf, tb = timed(fwd), timed(bwd)
    print("p_in=%.1f  ln_fwd %6.1f us (%.2f TB/s)   ln_bwd %6.1f us (%.2f TB/s)" % (
        p, tf, 4 * M * H * 2 / tf / 1e6, tb, 4 * M * H * 2 / tb / 1e6))
