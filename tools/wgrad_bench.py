#!/usr/bin/env python
"""grouped weight-gradient kernel on the GPU box: the four Linears of a BERT-base layer in one launch, time vs tokens
usage: python tools/wgrad_bench.py [iters]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visualbert_amd import _lib, ops
if os.environ.get("VB_DEV") == "1":
    _lib.use_dev_library()
dev = torch.device("cuda", 0)
L = _lib.lib()
TOK = [int(x) for x in os.environ.get('VB_TOKENS', '20992,41984,83968').split(',')]
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
H, I = 768, 3072
shapes = [(H, I), (I, H), (H, H), (3 * H, H)]            # (out, in): FFN-out, FFN-in, attention-out, QKV
g = torch.Generator().manual_seed(0)
def run(tokens, group):
    dys = [(torch.randn(tokens, o, generator=g) * 0.1).to(torch.bfloat16).to(dev) for o, _ in shapes]
    xs = [(torch.randn(tokens, i, generator=g) * 0.5).to(torch.bfloat16).to(dev) for _, i in shapes]
    dws = [torch.zeros(o, i, device=dev) for o, i in shapes]
    n = len(shapes)
    PA, I64, I32 = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int * n
    def grouped():
        rc = L.vb_wgrad_grouped(_lib.VB_BF16, n, PA(*[_lib.ptr(t) for t in dys]), I64(*[t.stride(0) for t in dys]),
                                PA(*[_lib.ptr(t) for t in xs]), I64(*[t.stride(0) for t in xs]),
                                PA(*[_lib.ptr(t) for t in dws]), I64(*[t.stride(0) for t in dws]),
                                I32(*[o for o, _ in shapes]), I32(*[i for _, i in shapes]), tokens, 1.0, None, _lib.stream_ptr())
        assert rc == 0
    def single():
        for dy, x, dw in zip(dys, xs, dws):
            ops.linear_wgrad(dy, x, dw)
    fn = grouped if group else single
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = sum(2.0 * tokens * o * i for o, i in shapes)
    # correctness of the last problem against fp32 (dW holds iters + 2 accumulations)
    ref = (dys[3].float().t() @ xs[3].float()) * (iters + 2)
    err = (dws[3] - ref).abs().max().item() / ref.abs().max().item()
    return ms * 1e3, fl / ms / 1e9, err
for tokens in TOK:
    # helper layout (default at 256 workgroups) against the classic two-slice layout (216 workgroups = 216 items), A/B/A/B
    res = {"helper": [], "classic": []}
    for rep in range(2):
        for name, wgs in (("helper", 0), ("classic", 216)):
            _lib.set_opts(persistent_workgroups=wgs)
            res[name].append(run(tokens, True))
    _lib.set_opts()
    h = min(res["helper"]); c = min(res["classic"])
    print("tokens %6d | helper layout %7.1f us %5.0f TF (relerr %.1e) | classic 216 items %7.1f us %5.0f TF (relerr %.1e)" % (
        tokens, h[0], h[1], h[2], c[0], c[1], c[2]))
