#!/usr/bin/env python
"""Would the fp8-cross-term mode also carry the BACKWARD pass?  CPU experiment on the oracle (test
infrastructure): BERT-base VisualBERT pre-training step, every nn.Linear replaced by an autograd function whose three products
    forward  y  = x . w^T        dgrad  dx = dy . w        wgrad  dw = dy^T . x
are each formed as  hi_a.hi_b + q(lo_a).q(hi_b) + q(hi_a).q(lo_b)  with hi = bf16(v), lo = v - hi and q = the arm's cross-term format:
    "bf16"      lo / hi as bf16 (what the shipping bf16x3 kernels compute)
    "e4m3 row"  real e4m3 planes (4 significant bits, range-limited) with ONE power-of-two scale per vector along the reduction axis'
                orthogonal index (a row of x / w for the forward, a row of dy / a column of w for the dgrad, a column of dy / x for the wgrad)
Reported: max |dlogit| of the forward and the relative L2 error of every parameter gradient against the fp32 step (worst, median).
The GPU suite bounds the bf16x3 gradients of this model at rel-L2 <= 1.5e-4 (tests/test_parity_at_scale.py).

    python tools/x3f8_grad_bits.py [--batch 2] > profiles/r04_x3f8_grad_bits.txt"""
import argparse
import os
import sys
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402
from oracle import visualbert_oracle as vo  # noqa: E402
from x3_cross_term_bits import bf16, e4m3_scaled  # noqa: E402


def planes(v, fmt, weight):
    """(hi, q(hi), q(lo)) of a [rows, K] matrix whose K is the reduction axis.  fmt "e4m3t": ONE scale per tensor (from a bound 2^4 too
    large) for activations and gradients, per-row scales for weights (their split pass owns whole rows)"""
    hi = bf16(v)
    lo = v - hi
    if fmt == "bf16":
        return hi, hi, bf16(lo)
    if fmt == "e4m3t" and not weight:
        return hi, e4m3_scaled(hi, -1, 4), e4m3_scaled(lo, -1, 4)
    return hi, e4m3_scaled(hi, 0), e4m3_scaled(lo, 0)


def prod(a, b, fmt, b_is_weight):
    """a [M, K] . b [N, K]^T with split operands"""
    ah, aq, al = planes(a, fmt, False)
    bh, bq, bl = planes(b, fmt, b_is_weight)
    return ah @ bh.t() + al @ bq.t() + aq @ bl.t()


def make_linear(fmt, backward_too):
    class Lin(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            ctx.save_for_backward(x, w)
            return prod(x.reshape(-1, x.size(-1)), w, fmt, True).reshape(*x.shape[:-1], w.size(0))

        @staticmethod
        def backward(ctx, dy):
            x, w = ctx.saved_tensors
            x2, dy2 = x.reshape(-1, x.size(-1)), dy.reshape(-1, dy.size(-1))
            f = fmt if backward_too else "exact"
            if f == "exact":
                return (dy2 @ w).reshape(x.shape), dy2.t() @ x2
            dx = prod(dy2, w.t().contiguous(), f, True)            # reduce over out-features: rows of dy, rows of W^T
            dw = prod(dy2.t().contiguous(), x2.t().contiguous(), f, False)   # reduce over tokens: rows of dy^T, rows of x^T
            return dx.reshape(x.shape), dw

    def linear(x, w, b, mode, part="enc"):
        y = Lin.apply(x, w)
        return y if b is None else y + b
    return linear


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--seed", type=int, default=11)
    args = ap.parse_args()
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    cfg = vo.OracleConfig(**vo.CONFIGS["base"])
    sd = vo.synth_state_dict(cfg, "pretraining", args.seed)
    batch = vo.synth_batch(cfg, args.batch, 128, 36, args.seed, "pretraining", ragged=True)

    def step():
        leaves = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in sd.items())
        out = vo.objective_forward(leaves, cfg, "pretraining", mode="fp32", **batch)
        out["loss"].mean().backward()
        return out["logits"].detach(), OrderedDict((k, v.grad) for k, v in leaves.items() if v.grad is not None)

    print("# BERT-base VisualBERT pre-training step (oracle, eval-mode dropout), B = %d x (128 tok + 36 regions), ragged; every nn.Linear's forward, "
          "dgrad and wgrad as split-operand products; reference = the fp32 step" % args.batch)
    ref_logits, ref_g = step()
    orig = vo.linear
    for name, fmt, bwd in (("cross terms bf16 (the shipping bf16x3 arithmetic), forward + backward", "bf16", True),
                           ("cross terms e4m3 / per-row scale, forward only (backward exact)", "e4m3", False),
                           ("cross terms e4m3 / per-row scale, forward + backward", "e4m3", True),
                           ("cross terms e4m3 / ONE scale per activation or gradient tensor (2^4 headroom), forward + backward", "e4m3t", True)):
        vo.linear = make_linear(fmt, bwd)
        try:
            lg, g = step()
        finally:
            vo.linear = orig
        # (the key biases are left out: softmax is invariant to them, their exact gradient is zero and the fp32 one is rounding noise)
        rel = sorted((float((g[k] - ref_g[k]).norm() / ref_g[k].norm().clamp_min(1e-30)), k) for k in ref_g
                     if float(ref_g[k].norm()) > 0 and not k.endswith("attention.self.key.bias"))
        vals = [r for r, _ in rel]
        big = [(r, k) for r, k in rel if ref_g[k].numel() >= 4096]         # matrices and embedding tables (biases of 2 .. 3072 elements apart)
        print("%-78s max|dlogit| %.2e   grad rel-L2: median %.2e  best %.2e  worst %.2e (%s)  worst of the >= 4096-element tensors %.2e (%s)" % (
            name, float((lg - ref_logits).abs().max()), vals[len(vals) // 2], vals[0], vals[-1], rel[-1][1], big[-1][0], big[-1][1]), flush=True)


if __name__ == "__main__":
    main()
