#!/usr/bin/env python
"""Where the HOST spends a training step at small per-GPU batches (round 6: with the small-token weight gradients and the split-K
GEMMs the kernels of a B = 8 step add up to 4.7 ms while the step takes 5.05 ms -- the step is host-bound there).
cProfile over N steps with the GPU kept behind (no synchronisation inside the loop); top functions by own and cumulative time.

    python tools/host_profile.py [--batch 8] [--steps 40] > profiles/r06_host_profile_b8.txt"""
import argparse
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=40)
    args = ap.parse_args()
    from visualbert_amd.data import synthetic_batch
    from visualbert_amd.model import AttrDict, ModelWrapper, VisualBERTFixedImageEmbedding
    from visualbert_amd.modeling import BertConfig
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    config = BertConfig(30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072)
    model = VisualBERTFixedImageEmbedding(config=config, training_head_type="pretraining", visual_embedding_dim=2048,
                                          compute_dtype=torch.bfloat16).to(dev)
    model.train()
    B = args.batch
    mw = ModelWrapper(AttrDict(train_batch_size=B, learning_rate=5e-5, warmup_proportion=0.1, num_train_epochs=1,
                               gradient_accumulation_steps=1), 1000 * B, model=model)
    batch = synthetic_batch("pretraining", B, 128, 36, 2048, 30522, seed=0, device=dev)
    for _ in range(8):
        mw.step(batch)
    torch.cuda.synchronize()
    # (1) plain wall time per step, host enqueue time per step (time until the loop returns) and the GPU's tail
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mw.step(batch)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("# B = %d, %d steps: host enqueue %.3f ms/step, wall %.3f ms/step (GPU tail after the loop %.3f ms in all)" % (
        B, args.steps, t_host / args.steps * 1e3, t_all / args.steps * 1e3, (t_all - t_host) * 1e3))
    # (2) cProfile of the same loop (the profiler itself adds ~30-50 %: read the SHARES)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(args.steps):
        mw.step(batch)
    pr.disable()
    torch.cuda.synchronize()
    for key in ("tottime", "cumulative"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(45)
        print("# ---- sorted by %s (%d steps) ----" % (key, args.steps))
        print("\n".join(l[:200] for l in s.getvalue().splitlines()[4:]))


if __name__ == "__main__":
    main()
