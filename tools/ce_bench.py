#!/usr/bin/env python
"""MLM cross-entropy kernel on the GPU box at the BERT-base shape: dense rows vs the compact labelled-row form
usage: python tools/ce_bench.py [batch]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visualbert_amd import _lib
dev = torch.device("cuda", 0)
L = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
M, V = B * 164, 30522
ld = (V + 63) // 64 * 64
g = torch.Generator().manual_seed(0)
logits = torch.randn(M, ld, device=dev)
lab = torch.full((M,), -1, dtype=torch.int64)
sel = torch.rand(M, generator=g) < 0.117
lab[sel] = torch.randint(0, V, (int(sel.sum()),), generator=g)
lab = lab.to(dev)
rows = torch.nonzero(lab != -1).reshape(-1)
n = rows.numel(); n_pad = (n + 63) // 64 * 64
acc = torch.empty(66, device=dev); loss = torch.empty(1, device=dev)
dl = torch.empty(M, ld, dtype=torch.bfloat16, device=dev)
dlc = torch.empty(n_pad, ld, dtype=torch.bfloat16, device=dev)
def bench(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
dense = lambda: L.vb_ce_fwd_bwd(_lib.VB_BF16, _lib.ptr(logits), ld, _lib.ptr(lab), -1, _lib.ptr(acc), _lib.ptr(loss), _lib.ptr(dl), ld, M, V, _lib.stream_ptr())
comp = lambda: L.vb_ce_fwd_bwd_rows(_lib.VB_BF16, _lib.ptr(logits), ld, _lib.ptr(lab), -1, _lib.ptr(rows), n, n_pad, _lib.ptr(acc), _lib.ptr(loss), _lib.ptr(dlc), ld, M, V, _lib.stream_ptr())
td, tc = bench(dense), bench(comp)
rb = n * V * 4
print("M=%d labelled rows %d: dense %.1f us | compact %.1f us (reads %.0f MB, writes %.0f MB -> %.2f TB/s)" % (M, n, td, tc, rb / 1e6, n_pad * ld * 2 / 1e6, (rb + n_pad * ld * 2) / tc / 1e6))
