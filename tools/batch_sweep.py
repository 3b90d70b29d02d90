#!/usr/bin/env python
"""Batch sweep of the pre-training step: the reference's configs train global batch 48-64
(configs/vqa/coco-pre-train.json:17), i.e. 6-8 samples per GPU at N = 8 -- the small-batch regime bench.py's per-GPU batch 1024
says nothing about.  Same step as bench.py (MLM + ITM, dropout on, dense decoder, BertAdam), B = 8 / 64 / 256 / 1024 per GPU, bf16 and
the split-operand bf16x3 mode; samples/s = B / median step time (HIP events around each step, no host sync inside the loop).

    python tools/batch_sweep.py [--batches 8 64 256 1024] [--dtypes bf16 bf16x3] > profiles/r05_batch_sweep.txt"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402


def run(dtype_name, B, steps, warmup, dev):
    from visualbert_amd.data import synthetic_batch
    from visualbert_amd.model import AttrDict, ModelWrapper, VisualBERTFixedImageEmbedding
    from visualbert_amd.modeling import BertConfig
    torch.manual_seed(1234)
    config = BertConfig(30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072)
    model = VisualBERTFixedImageEmbedding(config=config, training_head_type="pretraining", visual_embedding_dim=2048,
                                          compute_dtype=bench.compute_dtype_of(dtype_name)).to(dev)
    model.train()
    mw = ModelWrapper(AttrDict(train_batch_size=B, learning_rate=5e-5, warmup_proportion=0.1, num_train_epochs=1,
                               gradient_accumulation_steps=1), (steps + warmup + 20) * B, model=model)
    batch = synthetic_batch("pretraining", B, 128, 36, 2048, 30522, seed=0, device=dev)
    for _ in range(warmup):
        mw.step(batch)
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    marks[0].record()
    for i in range(steps):
        mw.step(batch)
        marks[i + 1].record()
    torch.cuda.synchronize()
    ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    med = ms[len(ms) // 2]
    del mw, model, batch
    torch.cuda.empty_cache()
    return med, ms[0], ms[-1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, nargs="+", default=[8, 16, 32, 64, 128, 256, 512, 1024])
    ap.add_argument("--dtypes", nargs="+", default=["bf16", "bf16x3"])
    ap.add_argument("--x3-batches", type=int, nargs="+", default=[8, 64, 256, 1024], help="the (slower) split-operand mode runs this subset")
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    fps = bench.flops_per_sample(12, 768, 3072, 30522, 164, 36, 2048, "pretraining")
    print("# pre-training step (BERT-base, 36 regions x 2048-d + 128 tokens, MLM + ITM, dropout on, dense decoder, BertAdam), 1 x MI355X;")
    print("# per-GPU batch sweep: samples/s = B / median of %d steps after %d warm-up; step_mfu = samples/s x %.2f GF / 2.5 PF/s" % (
        args.steps, args.warmup, fps / 1e9))
    print("%-8s %6s %12s %12s %12s %10s" % ("dtype", "B", "median ms", "min ms", "samples/s", "step_mfu"))
    for dt in args.dtypes:
        for B in (args.x3_batches if dt == "bf16x3" else args.batches):
            try:
                med, lo, hi = run(dt, B, args.steps, args.warmup, dev)
                print("%-8s %6d %12.3f %12.3f %12.1f %10.4f" % (dt, B, med, lo, B / med * 1e3, B / med * 1e3 * fps / 2.5e15), flush=True)
            except RuntimeError as e:
                print("%-8s %6d  failed: %s" % (dt, B, str(e)[:120]), flush=True)


if __name__ == "__main__":
    main()
