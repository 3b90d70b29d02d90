#!/usr/bin/env python
"""VERDICT r03 "Next" 1(d): do the two CROSS terms of the split-operand product (lo.hi and hi.lo, 2^-9 below hi.hi) survive being
rounded to fewer bits -- i.e. could they ride a cheaper matrix pipe (fp8 MFMA runs at twice the bf16 rate on gfx950)?

CPU experiment on the oracle (test infrastructure), BERT-base VisualBERT pre-training forward, BASELINE configs[1] shapes: every
nn.Linear of the model is replaced by   y = hi_x.hi_w + q_lo(lo_x).q_hi(hi_w) + q_hi(hi_x).q_lo(lo_w)   with hi = bf16(v), lo = v - hi
and q_n = round-to-nearest-even to n significant bits (8 = bf16 = what the kernels do; 4 = fp8 e4m3; 3 = fp8 e5m2; exponent range
is NOT limited here, i.e. perfect per-block scaling is assumed -- an optimistic bound for an fp8 pipe).  The attention core stays
exact.  max|dlogit| against the fp32 forward is what the north-star bounds at 1e-3.

    python tools/x3_cross_term_bits.py [--batch 2] > profiles/r04_x3_cross_term_bits.txt
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
from oracle import visualbert_oracle as vo  # noqa: E402


def bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def rnd(x, bits):
    if bits >= 24:
        return x
    m, e = torch.frexp(x)
    return torch.ldexp(torch.round(m * (1 << bits)) / (1 << bits), e)


def make_linear(bits_hi, bits_lo, cross=True):
    def linear(x, w, b, mode, part="enc"):
        xh, wh = bf16(x), bf16(w)
        y = F.linear(xh, wh)
        if cross:
            xl, wl = x - xh, w - wh
            y = y + F.linear(rnd(xl, bits_lo), rnd(wh, bits_hi)) + F.linear(rnd(xh, bits_hi), rnd(wl, bits_lo))
        return y if b is None else y + b
    return linear


def e4m3_scaled(x, block, slack=0):
    """what an fp8 plane would really hold: x -> OCP e4m3 (4 significant bits, normal exponents 2^-6 .. 2^8, subnormal spacing 2^-9,
    max 448) under a power-of-two scale per row (block = 0) or per `block` consecutive K elements of a row, chosen so that the
    group's largest magnitude lands in [224, 448]; returns the dequantised values"""
    K = x.size(-1)
    g = x.reshape(1, -1) if block < 0 else (x.reshape(-1, K) if block == 0 else x.reshape(-1, block))     # block < 0: ONE scale per tensor
    amax = g.abs().amax(-1, keepdim=True)
    # (slack: the scale is chosen from an a-priori BOUND of the row's magnitude that is 2^slack too large -- what a producer that does not
    #  own whole rows would have to do, e.g. |y_row| <= |x_row|_2 max_j |W_j|_2 for a GEMM epilogue: the plane uses 2^-slack of its range)
    sc = torch.exp2(torch.floor(torch.log2(448.0 / amax.clamp_min(2.0 ** -100))) - slack).clamp(max=2.0 ** 100)   # power of two: the E8M0 scale's reciprocal
    y = g * sc
    m, e = torch.frexp(y)                                            # y = m 2^e, |m| in [0.5, 1)
    e_eff = torch.clamp(e, min=-5)                                   # below 2^-6 the spacing stays 2^-9 (subnormals)
    q = torch.ldexp(torch.round(torch.ldexp(m, e - e_eff) * 16) / 16, e_eff)
    q = torch.clamp(q, -448.0, 448.0)
    return (q / sc).reshape(x.shape)


def make_linear_fp8(block, slack=0):
    def linear(x, w, b, mode, part="enc"):
        xh, wh = bf16(x), bf16(w)
        xl, wl = x - xh, w - wh
        # (the slack applies to the ACTIVATION planes only: weights are split by a pass that owns whole rows)
        wb = max(block, 0)
        y = F.linear(xh, wh) + F.linear(e4m3_scaled(xl, block, slack), e4m3_scaled(wh, wb)) + F.linear(e4m3_scaled(xh, block, slack), e4m3_scaled(wl, wb))
        return y if b is None else y + b
    return linear


ARMS = [("hi.hi only (plain bf16 operands in every Linear)", None),
        ("cross terms on bf16 operands: hi 8 bits, lo 8 bits  (= the bf16x3 kernels)", (8, 8)),
        ("cross terms: hi 6 bits, lo 6 bits", (6, 6)),
        ("cross terms: hi 5 bits, lo 5 bits", (5, 5)),
        ("cross terms on fp8 e4m3-like operands: hi 4 bits, lo 4 bits", (4, 4)),
        ("cross terms: hi 4 bits, lo 8 bits (only the LARGE operand of each cross term narrowed)", (4, 8)),
        ("cross terms on fp8 e5m2-like operands: hi 3 bits, lo 3 bits", (3, 3)),
        ("cross terms in REAL e4m3 (range-limited), one power-of-two scale per 32 K elements of a row", "fp8:32"),
        ("cross terms in REAL e4m3 (range-limited), ONE power-of-two scale per ROW (no scale traffic in the K loop)", "fp8:0"),
        ("  ... per-row scale from a bound 2^3 too large (activation planes)", "fp8:0:3"),
        ("  ... per-row scale from a bound 2^5 too large", "fp8:0:5"),
        ("  ... per-row scale from a bound 2^7 too large", "fp8:0:7"),
        ("  ... per-row scale from a bound 2^9 too large", "fp8:0:9"),
        ("cross terms in REAL e4m3, ONE scale per TENSOR for the activation planes (per row for the weights), exact tensor maximum", "fp8:-1:0"),
        ("  ... one scale per tensor from a bound 2^4 too large (e.g. last step's maximum with headroom)", "fp8:-1:4"),
        ("  ... one scale per tensor from a bound 2^8 too large", "fp8:-1:8")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--seeds", type=int, nargs="+", default=[11, 12])
    args = ap.parse_args()
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    cfg = vo.OracleConfig(**vo.CONFIGS["base"])
    print("# split-operand product with NARROWED cross terms: BERT-base VisualBERT pre-training forward, B=%d x (128 tok + 36 regions), "
          "ragged; every nn.Linear emulated, attention core exact; reference = the fp32 forward; north-star tolerance 1e-3" % args.batch)
    orig = vo.linear
    for seed in args.seeds:
        sd = vo.synth_state_dict(cfg, "pretraining", seed)
        batch = vo.synth_batch(cfg, args.batch, 128, 36, seed, "pretraining", ragged=True)
        with torch.no_grad():
            ref = vo.objective_forward(sd, cfg, "pretraining", mode="fp32", **batch)["logits"]
        print("\nseed %d: fp32 logits absmax %.3f" % (seed, float(ref.abs().max())))
        for name, bits in ARMS:
            vo.linear = (make_linear(0, 0, cross=False) if bits is None else make_linear_fp8(*[int(v) for v in bits.split(':')[1:]]) if isinstance(bits, str)
                         else make_linear(*bits))
            try:
                with torch.no_grad():
                    lg = vo.objective_forward(sd, cfg, "pretraining", mode="fp32", **batch)["logits"]
            finally:
                vo.linear = orig
            d = (lg - ref).abs()
            print("  %-92s max|dlogit| %.3e  mean %.3e  %s" % (name, float(d.max()), float(d.mean()),
                                                              "meets 1e-3" if float(d.max()) <= 1e-3 else "MISSES 1e-3"))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
