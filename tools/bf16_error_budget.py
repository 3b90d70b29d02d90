#!/usr/bin/env python
"""Where does the bf16 path's max|dlogit| against the fp32 reference come from?  

BERT-base VisualBERT pre-training forward, BASELINE configs[1] shapes (36 regions + 128 tokens, ragged), through the
oracle (test infrastructure) with the bf16 roundings of the HIP path switched on ONE GROUP AT A TIME
(oracle.visualbert_oracle.BF16_SITES).  Pure CPU: the attribution is a property of the arithmetic, not of the GPU;
tests/test_parity_at_scale.py measures the kernels themselves against the same fp32 reference on the device.

    python tools/bf16_error_budget.py [--batch 2] [--seeds 11 12] > profiles/r02_bf16_error_budget.txt
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
from oracle import visualbert_oracle as vo  # noqa: E402

ENC_STORE = "emb,qkv,probs,ctx,dense,ln"
ARMS = [
    ("everything bf16 (the HIP path's rounding points)", "bf16"),
    ("weights only (bf16 shadows, activations fp32)", "bf16:w_enc,w_head"),
    ("activations only (weights fp32)", "bf16:x_enc,x_head,head," + ENC_STORE),
    ("encoder bf16, head fp32", "bf16:w_enc,x_enc," + ENC_STORE),
    ("encoder fp32, head bf16 (transform + tied decoder)", "bf16:w_head,x_head,head"),
    ("tied decoder operands only: head weights", "bf16:w_head"),
    ("residual stream only (LayerNorm outputs rounded)", "bf16:ln"),
    ("everything bf16 EXCEPT LN outputs (fp32 in the residual add AND as GEMM input: not implementable on bf16 MFMA)",
     "bf16:w_enc,w_head,x_head,head,emb,qkv,probs,ctx,dense"),
    ("HYBRID A: fp32 residual stream (LN outputs fp32 in the residual add), bf16 GEMM operands everywhere",
     "bf16:w_enc,w_head,x_enc,x_head,head,emb,qkv,probs,ctx,dense"),
    ("HYBRID B: A + fp32 LayerNorm inputs (attention-out / FFN-out GEMMs write fp32)",
     "bf16:w_enc,w_head,x_enc,x_head,head,emb,qkv,probs,ctx"),
    ("HYBRID C: B + fp32 head (transform + decoder in fp32: 3-pass split-bf16 decoder equivalent)",
     "bf16:w_enc,x_enc,emb,qkv,probs,ctx"),
    ("attention internals only (q, k, v, P, context)", "bf16:qkv,probs,ctx"),
    ("everything bf16 EXCEPT attention internals", "bf16:w_enc,w_head,x_enc,x_head,head,emb,dense,ln"),
    ("dense / GELU outputs only", "bf16:dense"),
]


def stats(lg, ref, labels):
    d = (lg - ref).abs().flatten()
    top1 = (lg.argmax(-1) == ref.argmax(-1)).float().mean().item()
    k = max(1, int(d.numel() * 1e-3))
    p999 = d.topk(k).values[-1].item()
    return dict(max=d.max().item(), mean=d.mean().item(), p999=p999, top1=top1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--seeds", type=int, nargs="+", default=[11, 12])
    ap.add_argument("--layers", type=int, default=12)
    args = ap.parse_args()
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    cfg = vo.OracleConfig(**dict(vo.CONFIGS["base"], num_hidden_layers=args.layers))
    print("# bf16 error budget: BERT-base (%d layers) VisualBERT pre-training forward, B=%d x (128 tok + 36 regions), ragged"
          % (args.layers, args.batch))
    print("# each arm rounds to bf16 ONLY at the listed sites of oracle/visualbert_oracle.py (BF16_SITES); reference = "
          "the same oracle in fp32")
    print("# columns: max / mean / p99.9 |dlogit| over all [B,S,V] logits, top-1 agreement, |dloss|")
    for seed in args.seeds:
        sd = vo.synth_state_dict(cfg, "pretraining", seed)
        batch = vo.synth_batch(cfg, args.batch, 128, 36, seed, "pretraining", ragged=True)
        with torch.no_grad():
            ref = vo.objective_forward(sd, cfg, "pretraining", mode="fp32", **batch)
        print("\nseed %d: fp32 logits absmax %.3f, rms %.3f, loss %.4f"
              % (seed, ref["logits"].abs().max().item(), ref["logits"].pow(2).mean().sqrt().item(), float(ref["loss"])))
        for name, mode in ARMS:
            t0 = time.time()
            with torch.no_grad():
                out = vo.objective_forward(sd, cfg, "pretraining", mode=mode, **batch)
            st = stats(out["logits"], ref["logits"], None)
            print("  %-102s max %.3e  mean %.3e  p99.9 %.3e  top1 %.5f  |dloss| %.2e  (%.0fs)"
                  % (name, st["max"], st["mean"], st["p999"], st["top1"], abs(float(out["loss"]) - float(ref["loss"])),
                     time.time() - t0))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
