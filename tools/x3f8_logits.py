#!/usr/bin/env python
"""End-to-end accuracy of the proposed fp8-cross-term mode with the REAL instruction: the oracle's BERT-base
VisualBERT pre-training forward (CPU, test infrastructure) with EVERY nn.Linear routed through the developer-library prototype on the
GPU -- vb_split_f8 on both operands (e4m3 planes, one power-of-two scale per row), vb_gemm_x3f8 (hi.hi on the bf16 pipe, lo8.hi8 and
hi8.lo8 on v_mfma_scale_f32_16x16x128_f8f6f4) -- against the same forward in fp32.  The attention core, LayerNorm, GELU and softmax stay
exact, as in tools/x3_cross_term_bits.py, whose CPU emulation predicted max |dlogit| 6.3e-4 - 6.5e-4; the north-star bound is 1e-3.

    python tools/x3f8_logits.py [--config base] [--batch 2] > profiles/r04_x3f8_logits.txt"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from oracle import visualbert_oracle as vo  # noqa: E402
from visualbert_amd import _lib  # noqa: E402


def up(n, m):
    return (n + m - 1) // m * m


class Proto:
    def __init__(self, dev, L):
        self.dev, self.L, self.wcache, self.calls = dev, L, {}, 0

    def split(self, x, K):
        rows = x.size(0)
        img = torch.zeros(rows, 4 * K, dtype=torch.uint8, device=self.dev)
        s = [torch.zeros(up(rows, 64), dtype=torch.uint8, device=self.dev) for _ in range(2)]
        _lib.check(self.L.vb_split_f8(_lib.ptr(x), x.stride(0), _lib.ptr(img), 2 * K, rows, x.size(1), _lib.ptr(s[0]), _lib.ptr(s[1]),
                                      _lib.stream_ptr()), "vb_split_f8")
        return img, s

    def linear(self, x, w, b, mode, part="enc"):
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).float()
        M, K0 = x2.shape
        N = w.size(0)
        Mp, Np, K = up(M, 256), up(N, 256), up(K0, 128)
        xg = torch.zeros(Mp, K0, device=self.dev)
        xg[:M] = x2.to(self.dev)
        key = (w.data_ptr(), tuple(w.shape))
        if key not in self.wcache:
            wg = torch.zeros(Np, K0, device=self.dev)
            wg[:N] = w.float().to(self.dev)
            self.wcache[key] = self.split(wg, K)
        (wi, ws), (xi, xs) = self.wcache[key], self.split(xg, K)
        C = torch.empty(Mp, Np, device=self.dev)
        _lib.check(self.L.vb_gemm_x3f8(_lib.ptr(xi), 2 * K, _lib.ptr(wi), 2 * K, _lib.ptr(C), Np, Mp, Np, K, None, _lib.ptr(xs[0]), _lib.ptr(xs[1]),
                                       _lib.ptr(ws[0]), _lib.ptr(ws[1]), _lib.stream_ptr()), "vb_gemm_x3f8")
        self.calls += 1
        y = C[:M, :N].cpu()
        if b is not None:
            y = y + b
        return y.reshape(*shp[:-1], N)


def trained_state(dev, steps, batch_size):
    """weights after `steps` optimizer steps of the PRODUCT bf16 training step on bench.py's synthetic pre-training batch (same model seed,
    same learning-rate schedule as the default bench run): what bench.py's parity side batch sees (|logit|max ~ 12)."""
    from visualbert_amd.data import synthetic_batch
    from visualbert_amd.model import AttrDict, ModelWrapper, VisualBERTFixedImageEmbedding
    from visualbert_amd.modeling import BertConfig
    torch.manual_seed(1234)
    config = BertConfig(30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072)
    model = VisualBERTFixedImageEmbedding(config=config, training_head_type="pretraining", visual_embedding_dim=2048,
                                          compute_dtype=torch.bfloat16).to(dev)
    model.train()
    total = 50 * 3 + 10 + 20
    mw = ModelWrapper(AttrDict(train_batch_size=batch_size, learning_rate=5e-5, warmup_proportion=0.1, num_train_epochs=1,
                               gradient_accumulation_steps=1), total * batch_size, model=model)
    batch = synthetic_batch("pretraining", batch_size, 128, 36, 2048, 30522, seed=0, device=dev)
    for _ in range(steps):
        mw.step(batch)
    torch.cuda.synchronize()
    cfg = vo.OracleConfig(**vo.CONFIGS["base"])
    keep = vo.param_shapes(cfg, "pretraining")
    sd = {k: v.detach().float().cpu() for k, v in model.bert.state_dict().items() if k in keep}
    del mw, model, batch
    torch.cuda.empty_cache()
    return sd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="base")
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--seeds", type=int, nargs="+", default=[11])
    ap.add_argument("--cases", nargs="*", default=[],
                    help="gate inputs (VERDICT r04 item 3): golden stems of tests/golden (base_pretraining_b16, base_pretraining_stress_b8: weights "
                         "and batch of the REAL reference's golden) and / or `trained` (bench.py's side batch, B = 16 ragged, on the weights "
                         "after --trained-steps product steps)")
    ap.add_argument("--trained-steps", type=int, default=120)
    ap.add_argument("--trained-batch", type=int, default=1024)
    args = ap.parse_args()
    if os.environ.get("VB_EMU") == "1":
        _lib.set_library(os.path.join(ROOT, "tests", "hipemu", "libvisualbert_emu.so"), "cpu")      # logic check of this script only
        dev = torch.device("cpu")
        L = _lib.lib()
    else:
        dev = torch.device("cuda", 0)
        L = None
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    cfg = vo.OracleConfig(**vo.CONFIGS[args.config])
    T, R = (128, 36) if args.config == "base" else (12, 5)
    runs = []                                         # (label, state dict, batch, golden logits (strided) or None)
    for seed in ([] if args.cases else args.seeds):
        runs.append(("synth seed %d, B = %d ragged" % (seed, args.batch), vo.synth_state_dict(cfg, "pretraining", seed),
                     vo.synth_batch(cfg, args.batch, T, R, seed, "pretraining", ragged=True), None))
    for case in args.cases:
        if case == "trained":
            sd = trained_state(dev, args.trained_steps, args.trained_batch)      # the PRODUCT library trains; the prototype is bound after it
            runs.append(("bench side batch (seed 77, B = 16 ragged) on the weights after %d product bf16 steps at B = %d" % (
                args.trained_steps, args.trained_batch), sd, vo.synth_batch(cfg, 16, T, R, 77, "pretraining", ragged=True), None))
        else:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from golden_util import load_case, LOGIT_STRIDE
            c, head, sd, batch, g = load_case(case)
            runs.append(("golden %s (REAL reference)" % case, sd, batch, torch.as_tensor(g["logits_strided"])))
    if L is None:
        L = _lib.use_dev_library()
    for name in ("vb_split_f8", "vb_gemm_x3f8"):
        fn = getattr(L, name)
        fn.restype, fn.argtypes = _lib.DEV_SIGNATURES[name]
    print("# every nn.Linear of the %s VisualBERT pre-training forward through vb_split_f8 + vb_gemm_x3f8 on %s; reference = the fp32 oracle forward "
          "(and the REAL reference's stored logits for the golden cases); north-star tolerance 1e-3; pre-registered gate (DESIGN section 7): "
          "go only if max|dlogit| <= 7e-4 on every case" % (args.config, dev))
    orig = vo.linear
    worst = 0.0
    for label, sd, batch, gold in runs:
        with torch.no_grad():
            ref = vo.objective_forward(sd, cfg, "pretraining", mode="fp32", **batch)["logits"]
        proto = Proto(dev, L)
        vo.linear = proto.linear
        try:
            with torch.no_grad():
                lg = vo.objective_forward(sd, cfg, "pretraining", mode="fp32", **batch)["logits"]
        finally:
            vo.linear = orig
        d = (lg - ref).abs()
        worst = max(worst, float(d.max()))
        top1 = float((lg.argmax(-1) == ref.argmax(-1)).float().mean())
        extra = ""
        if gold is not None:
            extra = "  vs REAL reference (strided): max %.3e" % float((lg[:, :, ::LOGIT_STRIDE] - gold).abs().max())
        print("%s: %d Linear calls; fp32 logits absmax %.3f; max|dlogit| %.3e  mean %.3e  p99.9 %.3e  top-1 agreement %.4f%s  -> %s" % (
            label, proto.calls, float(ref.abs().max()), float(d.max()), float(d.mean()),
            float(d.flatten().topk(max(1, d.numel() // 1000)).values[-1]), top1, extra,
            "meets 1e-3" if float(d.max()) <= 1e-3 else "MISSES 1e-3"), flush=True)
    print("# worst case max|dlogit| %.3e -> gate (<= 7e-4 on all): %s" % (worst, "GO" if worst <= 7e-4 else "NO-GO"))


if __name__ == "__main__":
    main()
