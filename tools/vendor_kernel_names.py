#!/usr/bin/env python
"""Which hipBLASLt / Tensile kernels torch.matmul picks for the training step's GEMM shapes (tools only): run under
`rocprofv3 --kernel-trace --stats` and read the kernel names -- they spell out the macro tile (MT), the MFMA shape and wave
tile (MI / MIWT), the workgroup shape, whether operands go to LDS directly (DTL) and the prefetch depths (PGR / PLR).
usage: python tools/vendor_kernel_names.py [batch]"""
import sys
import torch
import torch.nn.functional as F

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
M = B * 164
dev = torch.device("cuda", 0)
for n, k in ((2304, 768), (768, 768), (3072, 768), (768, 3072), (768, 2304), (30522, 768)):
    a = torch.randn(M, k, device=dev).to(torch.bfloat16)
    w = torch.randn(n, k, device=dev).to(torch.bfloat16)
    b = torch.randn(n, device=dev).to(torch.bfloat16)
    for _ in range(3):
        F.linear(a, w, b)
        torch.matmul(a, w.t())
    torch.cuda.synchronize()
    print("M=%d N=%d K=%d done" % (M, n, k), flush=True)
    del a, w
