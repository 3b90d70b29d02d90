"""TEST INFRASTRUCTURE ONLY.  Generates tests/golden/*.npz from the REAL reference.

Run in the build container (where /root/reference exists):

    python oracle/make_golden.py

The reference modules are imported in place (oracle/reference_shim.py); weights
and batches are the deterministic synthetic ones of oracle/visualbert_oracle.py
(synth_state_dict / synth_batch), so a test on the GPU box can rebuild the same
inputs from (config name, head, seed) alone and compare against the stored
reference outputs.  Only small slices / checksums of large tensors are stored.

What runs is the reference's own code:
  TrainVisualBERTObjective.forward      pytorch_pretrained_bert/modeling.py:1373
  BertAdam.step                         pytorch_pretrained_bert/optimization.py:239
driven by a restatement of ModelWrapper.step (models/model_wrapper.py:64-96) and of
the image_mask construction (models/model.py:262-268), because allennlp (needed by
models/*.py) is not installable here.
"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import visualbert_oracle as vo          # noqa: E402
from oracle.reference_shim import load_reference, cpu_cuda_noop  # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(HERE), "tests", "golden")

# (file stem, config name, head, B, T, R, seed[, options]); options: bypass=True -> bypass_transformer
# (modeling.py:1299-1314), alignment=A -> image_text_alignment [B,R,A] (modeling.py:1223-1245)
CASES = [
    ("tiny_pretraining", "tiny", "pretraining", 2, 32, 8, 0),     # BASELINE.json configs[0]
    ("micro_pretraining", "micro", "pretraining", 3, 12, 5, 1),   # odd sizes: ragged tiles everywhere
    ("micro_vqa", "micro", "vqa", 3, 10, 6, 2),
    ("micro_nlvr", "micro", "nlvr", 2, 12, 8, 3),
    # the branches and heads outside BASELINE.json's configs
    ("micro_bypass", "micro", "pretraining", 3, 12, 5, 4, dict(bypass=True)),
    ("micro_align", "micro", "pretraining", 3, 12, 5, 5, dict(alignment=3)),
    ("micro_multichoice", "micro", "multichoice", 2, 10, 6, 6),
    ("micro_vqa_advanced", "micro", "vqa_advanced", 3, 12, 5, 7),
    ("micro_flickr", "micro", "flickr", 3, 12, 6, 8),
    ("micro_textonly", "micro", "pretraining", 3, 12, 5, 10, dict(text_only=True)),   # image_feat_variable = None
    # BASELINE.json configs[1], [3], [4] at their REAL size (BERT-base 12L/768), batch large enough that the token-major
    # GEMMs span several 256-row tiles.  compact=True: large tensors are stored as strided sub-samples (SUB_MAX elements)
    ("base_pretraining_b16", "base", "pretraining", 16, 128, 36, 31, dict(compact=True)),   # S = 164, M = 2624 tokens
    ("base_vqa_b16", "base", "vqa", 16, 20, 36, 32, dict(compact=True)),                    # S = 56
    ("base_nlvr_b8", "base", "nlvr", 8, 40, 72, 33, dict(compact=True)),                    # S = 112 (2 x 36 regions)
    # trained-like stress weights (oracle.stress_state_dict): wide attention scores, LayerNorm gamma outliers, channels with
    # |beta| > 2 |gamma| in the odd layers, outlier embedding rows, bias sigma 0.2 -- what a checkpoint loaded through
    # modeling.py:486-596 looks like and an init-distribution model does not
    ("base_pretraining_stress_b8", "base", "pretraining", 8, 128, 36, 34, dict(compact=True, stress=True)),
]

LR, WARMUP, T_TOTAL = 5e-5, 0.1, 100
LOGIT_STRIDE = 509
N_STEPS = 3
SUB_MAX = 1024          # compact cases: at most this many strided elements of a large tensor are stored


def sub(t):
    """strided sub-sample of a tensor (the same rule on both sides of a comparison: tests/golden_util.sub)."""
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // SUB_MAX)
    return f[::step][:SUB_MAX].float().numpy().copy()


def build_reference_model(cfg_kwargs, head, sd, bypass=False):
    ref_modeling, _ = load_reference()
    kw = dict(cfg_kwargs)
    vdim = kw.pop("visual_embedding_dim")
    V = kw.pop("vocab_size")
    config = ref_modeling.BertConfig(V, **kw)
    model = ref_modeling.TrainVisualBERTObjective(config, head, visual_embedding_dim=vdim,
                                                  bypass_transformer=bypass)
    model.bert.embeddings.special_intialize()
    missing = model.load_state_dict(sd, strict=False)
    # the only key we do not supply is the tied decoder weight alias
    assert set(missing.missing_keys) <= {"cls.predictions.decoder.weight"}, missing
    assert not missing.unexpected_keys, missing
    if head in ("pretraining", "vqa_advanced", "flickr"):
        assert model.cls.predictions.decoder.weight is model.bert.embeddings.word_embeddings.weight
    return model


def reference_forward(model, batch):
    """Restates models/model.py:262-288 (mask construction + kwargs mapping), then calls the reference."""
    feats = batch.get("image_feat_variable")
    image_mask = vo.build_image_mask(feats, batch["image_dim_variable"]) if feats is not None else None   # model.py:269-270
    with cpu_cuda_noop():
        return model(input_ids=batch["bert_input_ids"], token_type_ids=batch["bert_input_type_ids"],
                     input_mask=batch["bert_input_mask"], visual_embeddings=feats,
                     position_embeddings_visual=None, image_mask=image_mask,
                     visual_embeddings_type=batch.get("visual_embeddings_type"),
                     image_text_alignment=batch.get("image_text_alignment"),
                     label=batch.get("label"), flickr_position=batch.get("flickr_position"),
                     masked_lm_labels=batch.get("masked_lm_labels"), is_random_next=batch.get("is_random_next"),
                     output_all_encoded_layers=False)


def reference_optimizer(model):
    """models/model_wrapper.py:100-139 restated (param groups, BertAdam)."""
    _, ref_opt = load_reference()
    named = [n for n in model.named_parameters() if "pooler" not in n[0]]
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    groups = [
        {"params": [p for n, p in named if not any(nd in n for nd in no_decay)], "weight_decay": 0.01},
        {"params": [p for n, p in named if any(nd in n for nd in no_decay)], "weight_decay": 0.0},
    ]
    return ref_opt.BertAdam(groups, lr=LR, warmup=WARMUP, t_total=T_TOTAL)


def head16(t):
    return t.detach().reshape(-1)[:16].double().numpy().copy()


def make_case(stem, cfg_name, head, B, T, R, seed, options=None):
    options = options or {}
    cfg_kwargs = vo.CONFIGS[cfg_name]
    cfg = vo.OracleConfig(bypass_transformer=bool(options.get("bypass")), **cfg_kwargs)
    sd = vo.stress_state_dict(cfg, head, seed) if options.get("stress") else vo.synth_state_dict(cfg, head, seed)
    batch = vo.synth_batch(cfg, B, T, R, seed, head, alignment=int(options.get("alignment", 0)))
    if options.get("text_only"):
        batch = OrderedDict((k, v) for k, v in batch.items() if not k.startswith("image_"))
    model = build_reference_model(cfg_kwargs, head, sd, bypass=cfg.bypass_transformer)
    rec = OrderedDict()
    rec["meta"] = np.array([B, T, R, seed], dtype=np.int64)

    # ---- eval-mode forward (dropout off): the logits parity target.  The encoder outputs are captured at the
    # BertVisualModel boundary (the bypass branch refuses output_all_encoded_layers=True, modeling.py:1300)
    model.eval()
    captured = []
    hook = model.bert.register_forward_hook(lambda m, i, o: captured.append(o))
    with torch.no_grad():
        out = reference_forward(model, batch)
    hook.remove()
    compact = bool(options.get("compact"))
    if compact:
        rec["sequence_output_sub"] = captured[0][0][:, :, ::13].numpy()
    else:
        rec["sequence_output"] = captured[0][0].numpy()
    rec["pooled_output"] = captured[0][1].numpy()
    rec["loss"] = out["loss"].double().numpy()
    if head == "vqa_advanced":
        rec["masked_lm_loss"] = out["masked_lm_loss"].double().numpy()
        rec["accuracy"] = np.float64(out["accuracy"])
        lg = out["logits"]
        rec["logits_strided"] = lg[:, :, ::LOGIT_STRIDE].numpy()
        rec["logits_argmax"] = lg.argmax(-1).numpy()
    elif head == "flickr":
        rec["accuracy"] = out["accuracy"].double().numpy()
        rec["upperbound_accuracy"] = out["upperbound_accuracy"].double().numpy()
        rec["entity_num"] = out["entity_num"].numpy()
    elif head == "pretraining":
        rec["masked_lm_loss"] = out["masked_lm_loss"].double().numpy()
        rec["next_sentence_loss"] = out["next_sentence_loss"].double().numpy()
        rec["seq_relationship_score"] = out["seq_relationship_score"].numpy()
        lg = out["logits"]
        rec["logits_strided"] = lg[:, :, ::LOGIT_STRIDE].numpy()
        rec["logits_absmax"] = lg.abs().max().double().numpy()
        rec["logits_sum"] = lg.double().sum().numpy()
        rec["logits_argmax"] = lg.argmax(-1).numpy()
    else:
        rec["logits"] = out["logits"].numpy()
        if head == "vqa":
            rec["accuracy"] = out["accuracy"].double().numpy()

    # ---- one full training step with dropout p=0 (gradient + optimizer parity)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    model.train()
    opt = reference_optimizer(model)
    opt.zero_grad()
    out = reference_forward(model, batch)
    loss = out["loss"].mean()
    loss.backward()
    rec["train_loss"] = loss.detach().double().numpy()
    gnames = []
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        gnames.append(n)
        rec["grad_norm/" + n] = p.grad.double().norm().numpy()
        rec["grad_head/" + n] = head16(p.grad)
        if compact:
            rec["grad_sub/" + n] = sub(p.grad)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        opt.step()
        # step 0 has LR multiplier 0 (optimization.py:166-167, progress 0 < warmup): run
        # N_STEPS-1 more full steps on the same batch so the stored weights really moved.
        for _ in range(N_STEPS - 1):
            opt.zero_grad()
            out = reference_forward(model, batch)
            out["loss"].mean().backward()
            opt.step()
    rec["final_loss"] = out["loss"].mean().detach().double().numpy()
    for n, p in model.named_parameters():
        rec["post_norm/" + n] = p.detach().double().norm().numpy()
        rec["post_head/" + n] = head16(p)
        rec["delta_norm/" + n] = (p.detach() - sd[n]).double().norm().numpy()
        if compact:
            rec["delta_sub/" + n] = sub(p.detach() - sd[n])
    rec["grad_names"] = np.array(gnames)
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    path = os.path.join(GOLDEN_DIR, stem + ".npz")
    np.savez_compressed(path, **rec)
    print("wrote %s (%d KB)" % (path, os.path.getsize(path) // 1024))


def make_attention_case(stem="micro_attention_weights", cfg_name="micro", B=2, T=12, R=5, seed=9):
    """output_attention_weights=True (modeling.py:1428-1442): the forward returns only the per-layer attention
    probabilities [B, nh, S, S] (eval mode -> no dropout on them) and loss None."""
    ref_modeling, _ = load_reference()
    cfg_kwargs = vo.CONFIGS[cfg_name]
    cfg = vo.OracleConfig(**cfg_kwargs)
    sd = vo.synth_state_dict(cfg, "pretraining", seed)
    batch = vo.synth_batch(cfg, B, T, R, seed, "pretraining")
    kw = dict(cfg_kwargs)
    vdim = kw.pop("visual_embedding_dim")
    V = kw.pop("vocab_size")
    model = ref_modeling.TrainVisualBERTObjective(ref_modeling.BertConfig(V, **kw), "pretraining",
                                                  visual_embedding_dim=vdim, output_attention_weights=True)
    model.load_state_dict(sd, strict=False)
    model.eval()
    with torch.no_grad():
        out = reference_forward(model, batch)
    assert out["loss"] is None
    rec = OrderedDict(meta=np.array([B, T, R, seed], dtype=np.int64))
    for i, w in enumerate(out["attention_weights"]):
        rec["attention_weights/%d" % i] = w.numpy()
    path = os.path.join(GOLDEN_DIR, stem + ".npz")
    np.savez_compressed(path, **rec)
    print("wrote %s (%d KB)" % (path, os.path.getsize(path) // 1024))


def make_lxrt_case(stem="micro_lxrt", B=3, Tl=12, Rv=7, seed=12, feat_dim=256):
    """the sibling model's cross-modality blocks (unsupervised_visualbert/src/lxrt/modeling.py): one LXRTXLayer (:660-712)
    and a VisualFeatEncoder (:715-747) of the REAL reference on the oracle's synthetic weights / inputs -- outputs, and
    the gradients of every parameter and both inputs for the loss sum(lang_out * Wl) + sum(visn_out * Wv) (eval mode)."""
    from oracle.reference_shim import load_reference_lxrt
    lx = load_reference_lxrt()
    cfg_kwargs = dict(vo.CONFIGS["micro"])
    cfg = vo.OracleConfig(**cfg_kwargs)
    kw = dict(cfg_kwargs)
    kw.pop("visual_embedding_dim")
    V = kw.pop("vocab_size")
    rc = lx.BertConfig(V, **kw)
    lx.VISUAL_CONFIG.set_visual_dims(feat_dim, 4)
    layer = lx.LXRTXLayer(rc)
    enc = lx.VisualFeatEncoder(rc)
    sd = vo.lxrt_synth_state_dict(cfg, seed, feat_dim)
    missing = layer.load_state_dict({k: v for k, v in sd.items() if not k.startswith("enc.")}, strict=True)
    enc.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("enc.")}, strict=True)
    layer.eval()
    enc.eval()
    x = vo.lxrt_synth_inputs(cfg, B, Tl, Rv, seed, feat_dim)
    lang = x["lang"].clone().requires_grad_(True)
    feats = x["feats"].clone().requires_grad_(True)
    visn_in = enc((feats, x["boxes"]))                   # the visual stream enters through the feature encoder
    lo, vo_ = layer(lang, x["lang_ext_mask"], visn_in, x["visn_ext_mask"])
    g = torch.Generator().manual_seed(9000 + seed)
    wl, wv = torch.randn(lo.shape, generator=g), torch.randn(vo_.shape, generator=g)
    loss = (lo * wl).sum() + (vo_ * wv).sum()
    loss.backward()
    rec = OrderedDict(meta=np.array([B, Tl, Rv, seed, feat_dim], dtype=np.int64))
    rec["visn_encoded"] = visn_in.detach().numpy()
    rec["lang_out"] = lo.detach().numpy()
    rec["visn_out"] = vo_.detach().numpy()
    rec["loss"] = loss.detach().double().numpy()
    rec["grad_in/lang"] = lang.grad.numpy()
    rec["grad_in/feats"] = feats.grad.numpy()
    for n, p_ in list(layer.named_parameters()) + [("enc." + n, p_) for n, p_ in enc.named_parameters()]:
        rec["grad/" + n] = p_.grad.numpy()
    path = os.path.join(GOLDEN_DIR, stem + ".npz")
    np.savez_compressed(path, **rec)
    print("wrote %s (%d KB)" % (path, os.path.getsize(path) // 1024))


def make_lxrt_encoder_case(stem="base_lxrt_encoder", B=4, Tl=20, Rv=36, seed=21, feat_dim=2048, n_l=2, n_r=1, n_x=2):
    """LXRTEncoder (unsupervised_visualbert/src/lxrt/modeling.py:769-905) at BERT-base width, both of its forms, from the REAL
    reference's classes:
      style/  visualbert_style = True  -- the one form the reference's constructor can build (:784-799): visn_fc + n_l BertLayers
              over the concatenated sequence;
      lrx/    the l / r / x stack of forward :893-905.  The reference's constructor stops at `assert(0)` (:803-804) before
              it builds this form, so the module is assembled here around that assert -- nn.Module.__init__, then the very
              attributes the constructor would have set (:806-822), from the reference's own BertLayer / LXRTXLayer /
              VisualFeatEncoder classes -- and the reference's own forward is what runs.
    Compact golden: strided sub-samples (golden_util.sub) of outputs, input gradients and every parameter's gradient for the
    loss sum(lang_out * Wl) + sum(visn_out * Wv), eval mode."""
    from torch import nn
    from oracle.reference_shim import load_reference_lxrt
    from tests.golden_util import sub
    rec = OrderedDict(meta=np.array([B, Tl, Rv, seed, feat_dim, n_l, n_r, n_x], dtype=np.int64))
    cfg = vo.OracleConfig(**vo.CONFIGS["base"])
    x = vo.lxrt_synth_inputs(cfg, B, Tl, Rv, seed, feat_dim)
    g = torch.Generator().manual_seed(9100 + seed)
    wl, wv = torch.randn((B, Tl, cfg.hidden_size), generator=g), torch.randn((B, Rv, cfg.hidden_size), generator=g)
    for tag, style in (("style", True), ("lrx", False)):
        lx = load_reference_lxrt(l_layers=n_l, x_layers=n_x, r_layers=n_r, visualbert_style=style)
        lx.VISUAL_CONFIG.visualbert_style = style
        lx.VISUAL_CONFIG.l_layers, lx.VISUAL_CONFIG.x_layers, lx.VISUAL_CONFIG.r_layers = n_l, n_x, n_r
        lx.VISUAL_CONFIG.set_visual_dims(feat_dim, 4)
        kw = dict(vo.CONFIGS["base"])
        kw.pop("visual_embedding_dim")
        rc = lx.BertConfig(kw.pop("vocab_size"), **kw)
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            if style:
                enc = lx.LXRTEncoder(rc)
            else:
                enc = lx.LXRTEncoder.__new__(lx.LXRTEncoder)
                nn.Module.__init__(enc)
                enc.visn_fc = lx.VisualFeatEncoder(rc)
                enc.num_l_layers, enc.num_x_layers, enc.num_r_layers = n_l, n_x, n_r
                enc.multi_choice, enc.visualbert_style = 0, False
                enc.layer = nn.ModuleList([lx.BertLayer(rc) for _ in range(n_l)])
                enc.x_layers = nn.ModuleList([lx.LXRTXLayer(rc) for _ in range(n_x)])
                enc.r_layers = nn.ModuleList([lx.BertLayer(rc) for _ in range(n_r)])
                enc.config = rc
        shapes = {n: tuple(p_.shape) for n, p_ in enc.named_parameters()}
        sd = vo.synth_named(shapes, seed)
        enc.load_state_dict(sd, strict=True)
        enc.eval()
        lang = x["lang"].clone().requires_grad_(True)
        feats = x["feats"].clone().requires_grad_(True)
        lo, vo_ = enc(lang, x["lang_ext_mask"], (feats, x["boxes"]), x["visn_ext_mask"])
        loss = (lo * wl).sum() + (vo_ * wv).sum()
        loss.backward()
        rec[tag + "/names"] = np.array(sorted(shapes))
        rec[tag + "/shapes"] = np.array([",".join(str(d) for d in shapes[n]) for n in sorted(shapes)])
        rec[tag + "/lang_out_sub"] = sub(lo).numpy()
        rec[tag + "/visn_out_sub"] = sub(vo_).numpy()
        rec[tag + "/out_absmax"] = np.array(float(max(lo.abs().max(), vo_.abs().max())))
        rec[tag + "/loss"] = loss.detach().double().numpy()
        rec[tag + "/grad_in_lang_sub"] = sub(lang.grad).numpy()
        rec[tag + "/grad_in_feats_sub"] = sub(feats.grad).numpy()
        for n, p_ in enc.named_parameters():
            rec[tag + "/grad_sub/" + n] = sub(p_.grad).numpy()
            rec[tag + "/grad_norm/" + n] = np.array(float(p_.grad.double().norm()))
        import sys as _sys
        for m in [k for k in _sys.modules if k == "param" or k.startswith("lxrt")]:
            del _sys.modules[m]                               # the module reads its switches at import time: re-import per form
    path = os.path.join(GOLDEN_DIR, stem + ".npz")
    np.savez_compressed(path, **rec)
    print("wrote %s (%d KB)" % (path, os.path.getsize(path) // 1024))


def make_schedule_fixture():
    """learning-rate multipliers of EVERY schedule class of the reference (optimization.py:37-173) over a short run,
    evaluated by the reference's own classes -> tests/golden/schedules.json."""
    import json
    _, ref_opt = load_reference()
    t_total = 40
    cases = [("ConstantLR", {}), ("WarmupLinearSchedule", dict(warmup=0.1)), ("WarmupConstantSchedule", dict(warmup=0.25)),
             ("WarmupCosineSchedule", dict(warmup=0.1, cycles=0.5)),
             ("WarmupCosineWithHardRestartsSchedule", dict(warmup=0.1, cycles=3.0)),
             ("WarmupCosineWithWarmupRestartsSchedule", dict(warmup=0.05, cycles=4.0))]
    out = dict(t_total=t_total, cases=[])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for name, kw in cases:
            sch = getattr(ref_opt, name)(t_total=t_total, **kw)
            out["cases"].append(dict(schedule=name, kwargs=kw,
                                     lr=[float(sch.get_lr(s, nowarn=True)) for s in range(t_total + 4)]))
    path = os.path.join(GOLDEN_DIR, "schedules.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote %s" % path)


if __name__ == "__main__":
    torch.set_num_threads(8)
    only = set(sys.argv[1:])                     # optional: the stems to (re)generate
    for case in CASES:
        if not only or case[0] in only:
            make_case(*case)
    if not only or "micro_attention_weights" in only:
        make_attention_case()
    if not only or "schedules" in only:
        make_schedule_fixture()
    if not only or "micro_lxrt" in only:
        make_lxrt_case()
    if not only or "base_lxrt_encoder" in only:
        make_lxrt_encoder_case()
