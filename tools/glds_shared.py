#!/usr/bin/env python
"""LDS-direct streaming when every wave reads the SAME small window (64 KB .. 16 MB, depth 4): is there a hot-line penalty
in L2?  (No: profiles/r02_glds_stream.txt.)   python tools/glds_shared.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from visualbert_amd import _lib
dev = torch.device("cuda", 0)
import _knobs
L = _knobs.L
buf = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
sink = torch.empty(4096, device=dev)
iters = 4000
for span_kb in (64, 256, 1024, 4096, 16384):
    span = span_kb << 10
    row = []
    for blocks in (256, 512):
        depth = 4
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
        row.append("%d blocks: %5.2f TB/s (%4.0f GB/s/CU)" % (blocks, tb, tb * 1e3 / 256))
    print("window %6d KB, depth 4 | %s" % (span_kb, " | ".join(row)))
