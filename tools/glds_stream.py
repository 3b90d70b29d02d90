#!/usr/bin/env python
"""LDS-direct streaming ceiling per CU: python tools/glds_stream.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visualbert_amd import _lib
dev = torch.device("cuda", 0)
import _knobs
L = _knobs.L
buf = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
sink = torch.empty(4096, device=dev)
iters = 4000
for span_mb, label in ((16, "16 MB window (L2/MALL resident)"), (1024, "1 GB window (HBM)")):
    span = span_mb << 20
    for blocks in (256, 512):
        row = []
        for depth in (1, 2, 4, 8, 16):
            if blocks == 512 and depth == 16:
                continue
            for _ in range(2):
                L.vb_glds_stream(depth, _lib.ptr(buf), span, iters, blocks, _lib.ptr(sink), _lib.stream_ptr())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                L.vb_glds_stream(depth, _lib.ptr(buf), span, iters, blocks, _lib.ptr(sink), _lib.stream_ptr())
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            tb = blocks * 8 * iters * 1024 / ms / 1e9
            row.append("depth %2d: %5.2f TB/s (%4.0f GB/s/CU)" % (depth, tb, tb * 1e3 / 256))
        print("%s, %d blocks | %s" % (label, blocks, " | ".join(row)))
