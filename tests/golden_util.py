"""Shared helpers: load a golden case and compare a candidate against it."""
import os

import numpy as np
import torch

from oracle import visualbert_oracle as vo

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = {
    "tiny_pretraining": ("tiny", "pretraining"),
    "micro_pretraining": ("micro", "pretraining"),
    "micro_vqa": ("micro", "vqa"),
    "micro_nlvr": ("micro", "nlvr"),
    # SURVEY 8f / N4 (options as in oracle/make_golden.py)
    "micro_bypass": ("micro", "pretraining", dict(bypass=True)),
    "micro_align": ("micro", "pretraining", dict(alignment=3)),
    "micro_multichoice": ("micro", "multichoice"),
    "micro_vqa_advanced": ("micro", "vqa_advanced"),
    "micro_flickr": ("micro", "flickr"),
    "micro_textonly": ("micro", "pretraining", dict(text_only=True)),
}
# BASELINE.json configs[1], [3], [4] at BERT-base size (compact goldens: strided sub-samples of the large tensors);
# kept out of CASES because the micro-case tests iterate over it and read full tensors
BASE_CASES = {
    "base_pretraining_b16": ("base", "pretraining"),
    "base_vqa_b16": ("base", "vqa"),
    "base_nlvr_b8": ("base", "nlvr"),
}
# trained-like stress weights (oracle.stress_state_dict): its own table -- the bf16 bounds of BASE_CASES were
# measured at init-distribution weights
STRESS_CASES = {
    "base_pretraining_stress_b8": ("base", "pretraining", dict(stress=True)),
}
SUB_MAX = 1024


def sub(t):
    """the strided sub-sample rule of oracle/make_golden.py (compact cases)."""
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // SUB_MAX)
    return f[::step][:SUB_MAX].float()
LR, WARMUP, T_TOTAL = 5e-5, 0.1, 100
LOGIT_STRIDE = 509
N_STEPS = 3


def load_case(stem):
    table = CASES if stem in CASES else (BASE_CASES if stem in BASE_CASES else STRESS_CASES)
    cfg_name, head = table[stem][:2]
    options = table[stem][2] if len(table[stem]) > 2 else {}
    g = np.load(os.path.join(GOLDEN_DIR, stem + ".npz"), allow_pickle=False)
    B, T, R, seed = [int(x) for x in g["meta"]]
    cfg = vo.OracleConfig(bypass_transformer=bool(options.get("bypass")), **vo.CONFIGS[cfg_name])
    sd = vo.stress_state_dict(cfg, head, seed) if options.get("stress") else vo.synth_state_dict(cfg, head, seed)
    batch = vo.synth_batch(cfg, B, T, R, seed, head, alignment=int(options.get("alignment", 0)))
    if options.get("text_only"):
        batch = type(batch)((k, v) for k, v in batch.items() if not k.startswith("image_"))
    return cfg, head, sd, batch, g


def maxdiff(a, b):
    a = torch.as_tensor(np.asarray(a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    return float((a - b).abs().max())


def record(group, key, value, fname="parity_small.json"):
    """append a measured value to gpurun_out/<fname> (merged back by gpurun; copied to profiles/ when it is to be tracked):
    the bf16 bounds asserted in the tests are 1.5 x what these records show."""
    import json
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", fname)
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        data = {}
        if os.path.isfile(path):
            with open(path) as f:
                data = json.load(f)
        data.setdefault(group, {})[key] = value
        with open(path, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
    except (OSError, ValueError):
        pass
