#!/usr/bin/env python
"""MFMA issue-rate ceiling at real clocks: python tools/mfma_peak.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visualbert_amd import _lib
dev = torch.device("cuda", 0)
import _knobs
L = _knobs.L
for blocks in (64, 128, 256, 512):
    out = torch.empty(blocks * 512, device=dev)
    for kind, name in ((0, "16x16x32"), (1, "32x32x16"), (2, "16x16x32, operands changing every instruction")):
        iters = 20000
        for _ in range(2):
            L.vb_mfma_peak(kind, iters, blocks, _lib.ptr(out), _lib.stream_ptr())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            L.vb_mfma_peak(kind, iters, blocks, _lib.ptr(out), _lib.stream_ptr())
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        flops = blocks * 8 * iters * 524288.0
        print("blocks=%d %s: %.3f ms  %.0f TF/s" % (blocks, name, ms, flops / ms / 1e9))
