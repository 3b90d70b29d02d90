#!/usr/bin/env python
"""A/B of the K-contiguous GEMM kernels on the training step's shapes, alternating arms in one process (A/B/A/B):
    python tools/gemm_ab.py [batch] [kernel ids ...]        default: 512  81 90
Every forward and dgrad GEMM of an encoder layer with the epilogue the step uses, plus the MLM decoder."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visualbert_amd import _lib, ops
if os.environ.get("VB_DEV") == "1":
    _lib.use_dev_library()
elif os.environ.get("VB_LIB_PATH"):                     # A/B of two product builds: VB_LIB_PATH=tools/libvisualbert_hip_ab_<arm>.so
    _lib.set_library(os.path.abspath(os.environ["VB_LIB_PATH"]))
NOCHECK = os.environ.get("VB_NOCHECK") == "1"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
arms = sys.argv[2:] or ["81", "90"]            # "90:8" = kernel 90 with the tile walk in stripes of 8 column tiles (VB_DEV=1)


def select(arm):
    k, stripe, dbg = (arm.split(":") + ["", ""])[:3]       # "92::40" = raw developer debug bits 40 (VB_DEV=1)
    _lib.set_opts(nt_kernel=int(k))
    if os.environ.get("VB_DEV") == "1":
        _lib.dev_lib().vb_gemm_set_debug(((int(stripe) << 8) if stripe else 0) | (int(dbg) if dbg else 0))

dev = torch.device("cuda", 0)
M = B * 164
g = torch.Generator().manual_seed(0)
A0, GS, MA, ADD = _lib.VB_ACT_NONE, _lib.VB_ACT_GELU_SAVE_GRAD, _lib.VB_ACT_MUL_AUX, "add"
# name, N, K, fp32 out, epilogue
shapes = [("qkv fwd", 2304, 768, False, A0), ("attn-out fwd", 768, 768, False, A0), ("ffn-in fwd gelu+gelu'", 3072, 768, False, GS),
          ("ffn-out fwd", 768, 3072, False, A0), ("ffn-out dgrad x gelu' + colsum", 3072, 768, False, MA),
          ("ffn-in dgrad + addend", 768, 3072, False, ADD), ("attn-out dgrad", 768, 768, False, A0),
          ("qkv dgrad + addend", 768, 2304, False, ADD), ("decoder fwd f32", 30522, 768, True, A0)]
tot = {a: 0.0 for a in arms}
for name, n, k, f32, epi in shapes:
    a = (torch.randn(M, k, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    ld = (n + 7) // 8 * 8
    w = torch.zeros(ld, k, dtype=torch.bfloat16, device=dev)
    w[:n] = (torch.randn(n, k, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    bias = torch.randn(n, generator=g).to(dev)
    out = torch.empty(M, ld, dtype=torch.float32 if f32 else torch.bfloat16, device=dev)[:, :n]
    kw = dict(out=out, bias=bias)
    if epi == GS:
        kw.update(act=GS, aux_out=torch.empty(M, ld, dtype=torch.bfloat16, device=dev)[:, :n])
    elif epi == MA:
        kw.update(act=MA, aux_in=torch.randn(M, ld, device=dev).to(torch.bfloat16)[:, :n], colsum_out=torch.zeros(n, device=dev), bias=None)
    elif epi == ADD:
        kw.update(addend=torch.randn(M, ld, device=dev).to(torch.bfloat16)[:, :n], bias=None)
    res = {arm: [] for arm in arms}
    ref = None
    for rep in range(3):
        for arm in arms:
            select(arm)
            for _ in range(2):
                ops.gemm(a, w[:n], M, n, k, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.gemm(a, w[:n], M, n, k, **kw)
            e1.record(); torch.cuda.synchronize()
            res[arm].append(e0.elapsed_time(e1) / 5 * 1e3)
            if rep == 0:
                cur = out.float().clone()
                if ref is None:
                    ref = cur
                else:
                    d = (cur - ref).abs().max().item()
                    assert NOCHECK or d <= 2e-2 * max(1.0, ref.abs().max().item()), (name, arm, d)
    select("0")
    fl = 2.0 * M * n * k
    print("%-34s" % name + "   ".join("k%s %7.1f us (%6.1f TF)" % (arm, min(res[arm]), fl / min(res[arm]) / 1e6) for arm in arms), flush=True)
    for arm in arms:
        tot[arm] += min(res[arm]) * (1 if "decoder" in name else 12)
    del a, w, out, kw
print("per step (12 layers + decoder): " + "   ".join("k%s %.2f ms" % (arm, tot[arm] / 1e3) for arm in arms))
