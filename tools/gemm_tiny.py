import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visualbert_amd import _lib, ops
dev = torch.device("cuda", 0)
L = _lib.lib()
g = torch.Generator().manual_seed(0)
def bench(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ops.gemm_profile_start()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    d = list(ops.gemm_profile_stop().values())[0]
    return d["ms"] / d["launches"] * 1e3
L.vb_gemm_set_variant(42)
for (m, n, k) in [(256, 128, 64), (256, 128, 768), (256*64, 128, 64), (256*246, 128, 64), (10496, 768, 64), (10496, 768, 128), (10496, 3072, 64)]:
    for dbg in (0, 64):
        L.vb_gemm_set_debug(dbg)
        a = (torch.randn(m, k, generator=g) * 0.5).to(torch.bfloat16).to(dev)
        w = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16).to(dev)
        out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        print("M=%d N=%d K=%d dbg=%d: %.1f us" % (m, n, k, dbg, bench(lambda: ops.gemm(a, w, m, n, k, out=out))))
L.vb_gemm_set_debug(0)
