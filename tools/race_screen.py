#!/usr/bin/env python
"""race screen for the asynchronous LDS-copy kernels on the GPU box: many launches of the persistent NT kernel and the
grouped wgrad kernel over shapes with 1..N tiles per workgroup, every result compared with fp32 torch matmuls."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visualbert_amd import _lib, ops
dev = torch.device("cuda", 0)
import _knobs
L = _knobs.L
g = torch.Generator().manual_seed(123)
bad = 0
n_checks = 0
BENCH_ROWS = "--bench-rows" in sys.argv              # add bench.py's own M = 1024 x 164 rows (tests/test_bench_shape.py)
NT_SHAPES = [(20992, 768, 768), (20992, 2304, 768), (5248, 3072, 768), (20992, 768, 3072), (1300, 1000, 192), (4096, 4096, 1024)]
TN_GROUPS = [(20992, [(768, 3072), (3072, 768), (768, 768), (2304, 768)]), (2496, [(30522, 768)]), (4608, [(768, 2048), (264, 520)])]
if BENCH_ROWS:
    NT_SHAPES += [(167936, 768, 768), (167936, 2304, 768), (167936, 768, 3072)]
    TN_GROUPS += [(167936, [(768, 3072), (3072, 768), (768, 768), (2304, 768)])]
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 6):
    for (m, n, k) in NT_SHAPES:
        a = (torch.randn(m, k, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        w = (torch.randn(n, k, generator=g) * 0.1).to(torch.bfloat16).to(dev)
        ref = a.float() @ w.float().t()
        for variant, wgs in ((81, 0), (81, 64), (80, 0), (81, 24), (90, 0), (100, 0), (100, 24), (101, 0), (101, 24)):
            _knobs.variant(variant); _knobs.wgs(wgs)
            out = torch.empty(m, n, dtype=torch.float32, device=dev)
            ops.gemm(a, w, m, n, k, out=out)
            err = (out - ref).abs().max().item() / max(1.0, ref.abs().max().item())
            n_checks += 1
            if not err < 2e-3:
                bad += 1; print("NT MISMATCH", it, (m, n, k), variant, wgs, err)
    _knobs.variant(1); _knobs.wgs(0)
    for tokens, shapes in TN_GROUPS:
        dys = [ops.alloc2d(tokens, o, torch.bfloat16, dev) for o, _ in shapes]
        xs = [ops.alloc2d(tokens, i, torch.bfloat16, dev) for _, i in shapes]
        for t in dys + xs:
            t.copy_((torch.randn(t.shape, generator=g) * 0.3).to(torch.bfloat16))
        for wgs in (0, 40):
            _knobs.wgs(wgs)
            dws = [torch.zeros(o, i, device=dev) for o, i in shapes]
            n_ = len(shapes)
            PA, I64, I32 = ctypes.c_void_p * n_, ctypes.c_int64 * n_, ctypes.c_int * n_
            rc = L.vb_wgrad_grouped(_lib.VB_BF16, n_, PA(*[_lib.ptr(t) for t in dys]), I64(*[t.stride(0) for t in dys]),
                                    PA(*[_lib.ptr(t) for t in xs]), I64(*[t.stride(0) for t in xs]),
                                    PA(*[_lib.ptr(t) for t in dws]), I64(*[t.stride(0) for t in dws]),
                                    I32(*[o for o, _ in shapes]), I32(*[i for _, i in shapes]), tokens, 1.0, None, _lib.stream_ptr())
            assert rc == 0
            for dy, x, dw in zip(dys, xs, dws):
                ref = dy.float().t() @ x.float()
                err = (dw - ref).abs().max().item() / max(1.0, ref.abs().max().item())
                n_checks += 1
                if not err < 2e-3:
                    bad += 1; print("TN MISMATCH", it, tokens, tuple(dw.shape), wgs, err)
    _knobs.wgs(0)
print("race screen: %d checks, %d mismatches" % (n_checks, bad))
sys.exit(1 if bad else 0)
