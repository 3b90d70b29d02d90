#!/usr/bin/env python
"""Which torch (ATen / memcpy) operations a training step still launches next to the library's own kernels, and from which Python line
(round 6: ~30 launches of 3-5 us per step, 3 % of a B = 8 step).  torch.profiler over a few steps, grouped by operator and source line.

    python tools/step_aten_ops.py [--batch 8] > profiles/r06_step_aten_ops_b8.txt"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=4)
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
    for _ in range(5):
        mw.step(batch)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(args.steps):
            mw.step(batch)
        torch.cuda.synchronize()
    rows = {}
    for ev in prof.events():
        if not ev.name.startswith("aten::") and "Memcpy" not in ev.name and "Memset" not in ev.name:
            continue
        dt = getattr(ev, "device_time_total", 0) or getattr(ev, "cuda_time_total", 0)
        self_dt = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
        if not self_dt:
            continue
        where = "?"
        for fr in (ev.stack or []):
            if "visualbert_amd" in fr or "bench" in fr:
                where = fr.split("/")[-1]
                break
        k = (ev.name, where)
        r = rows.setdefault(k, [0, 0.0])
        r[0] += 1
        r[1] += self_dt
    print("# B = %d, %d profiled steps: torch operators with device time (per step: calls, self device us), by calling line" % (B, args.steps))
    tot = 0.0
    for (name, where), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print("%-34s %-70s %6.1f calls %8.1f us" % (name, where[:70], n / args.steps, us / args.steps))
        tot += us / args.steps
    print("# total %.1f us of torch-operator device time per step" % tot)


if __name__ == "__main__":
    main()
