#!/usr/bin/env python
"""which framework ops are behind the small device-to-device copies of a training step (torch profiler, GPU box)"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from visualbert_amd.model import ModelWrapper, AttrDict, VisualBERTFixedImageEmbedding
from visualbert_amd.modeling import BertConfig
from visualbert_amd.data import synthetic_pretraining_batch
dev = torch.device("cuda", 0)
B, T, R, Dv, V = 32, 128, 36, 2048, 30522
cfg = BertConfig(V)
model = VisualBERTFixedImageEmbedding(config=cfg, training_head_type="pretraining", visual_embedding_dim=Dv,
                                      compute_dtype=torch.bfloat16).to(dev)
args = AttrDict(dict(learning_rate=5e-5, warmup_proportion=0.1, num_train_epochs=1, train_batch_size=B, gradient_accumulation_steps=1))
mw = ModelWrapper(args, 1000, model=model)
batch = synthetic_pretraining_batch(B, T, R, Dv, V, seed=0, device=dev)
for _ in range(2): mw.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    mw.step(batch)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    n = e.name
    if "Memcpy" in n or "copyBuffer" in n or "hipMemcpy" in n:
        st = [s for s in (e.stack or []) if "visualbert_amd" in s or "bench" in s]
        cnt[(n[:40], st[0][-70:] if st else "")] += 1
for (n, st), c in cnt.most_common(25):
    print("%4d  %-40s %s" % (c, n, st))
