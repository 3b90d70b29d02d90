"""Whole-path parity of the HIP model against (a) golden outputs of the REAL reference
(tests/golden, fp32 mode, logits tolerance 1e-3 as BASELINE.json's north_star states -- measured
error is ~1e-5) and (b) the oracle restatement in bf16-emulation mode (bf16 kernels)."""
import copy

import numpy as np
import pytest
import torch

from oracle import visualbert_oracle as vo
from golden_util import CASES, LR, WARMUP, T_TOTAL, LOGIT_STRIDE, N_STEPS, load_case, maxdiff, record

pytestmark = pytest.mark.gpu


def build_model(cfg, head, sd, dev, dtype=torch.float32, dropout=None):
    from visualbert_amd.modeling import BertConfig
    from visualbert_amd.model import VisualBERTFixedImageEmbedding
    hd = cfg.hidden_dropout_prob if dropout is None else dropout
    ad = cfg.attention_probs_dropout_prob if dropout is None else dropout
    bc = BertConfig(cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                    num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                    hidden_dropout_prob=hd, attention_probs_dropout_prob=ad,
                    max_position_embeddings=cfg.max_position_embeddings, type_vocab_size=cfg.type_vocab_size)
    model = VisualBERTFixedImageEmbedding(config=bc, training_head_type=head,
                                          visual_embedding_dim=cfg.visual_embedding_dim, compute_dtype=dtype,
                                          bypass_transformer=getattr(cfg, "bypass_transformer", False))
    model = model.to(dev)
    own = model.bert.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            own[k].copy_(v)
    return model


def to_dev(batch, dev):
    return {k: v.to(dev) for k, v in batch.items()}


# the two kernel modes that must meet the north-star's tolerance: fp32 kernels (fp32-input MFMA) and "bf16x3" (fp32
# activations, every GEMM as three bf16 MFMA passes over hi / lo split operands; tests/test_bf16x3.py has the kernel tests)
STRICT_MODES = {"fp32": torch.float32, "bf16x3": "bf16x3"}


@pytest.mark.parametrize("mode", sorted(STRICT_MODES))
@pytest.mark.parametrize("stem", sorted(CASES))
def test_fp32_forward_matches_reference_golden(dev, stem, mode):
    cfg, head, sd, batch, g = load_case(stem)
    model = build_model(cfg, head, sd, dev, dtype=STRICT_MODES[mode])
    model.eval()
    captured = []           # encoder outputs at the BertVisualModel boundary (as oracle/make_golden.py captures them)
    hook = model.bert.bert.register_forward_hook(lambda m, i, o: captured.append(o))
    with torch.no_grad():
        out = model(**to_dev(batch, dev))
    hook.remove()
    assert maxdiff(captured[0][0].float().cpu(), g["sequence_output"]) < 1e-3
    assert maxdiff(captured[0][1].float().cpu(), g["pooled_output"]) < 1e-3
    assert abs(float(out["loss"].detach()) - float(g["loss"])) < 1e-3
    if head == "vqa_advanced":
        lg = out["logits"].float().cpu()
        assert maxdiff(lg[:, :, ::LOGIT_STRIDE], g["logits_strided"]) < 1e-4
        assert abs(float(out["masked_lm_loss"]) - float(g["masked_lm_loss"])) < 1e-4
        assert np.array_equal(lg.argmax(-1).numpy(), g["logits_argmax"])
        assert abs(float(out["accuracy"]) - float(g["accuracy"])) < 1e-9
    elif head == "flickr":
        for k in ("accuracy", "upperbound_accuracy", "entity_num"):     # integer counts: exact
            assert abs(float(out[k]) - float(g[k])) < 1e-6, k
    elif head == "pretraining":
        lg = out["logits"].float().cpu()
        err = maxdiff(lg[:, :, ::LOGIT_STRIDE], g["logits_strided"])
        assert err < 1e-3, err                                           # north_star: logits within 1e-3
        assert err < 1e-4, err                                           # what fp32 kernels actually deliver
        assert abs(float(lg.abs().max()) - float(g["logits_absmax"])) < 1e-3
        assert maxdiff(out["seq_relationship_score"].cpu(), g["seq_relationship_score"]) < 1e-4
        assert abs(float(out["masked_lm_loss"]) - float(g["masked_lm_loss"])) < 1e-4
        assert abs(float(out["next_sentence_loss"]) - float(g["next_sentence_loss"])) < 1e-4
        assert np.array_equal(lg.argmax(-1).numpy(), g["logits_argmax"])  # token indexing bit-exact
    else:
        assert maxdiff(out["logits"].float().cpu().reshape(g["logits"].shape), g["logits"]) < 1e-4
    if head == "vqa":
        assert abs(float(out["accuracy"]) - float(g["accuracy"])) < 1e-6


@pytest.mark.parametrize("mode", sorted(STRICT_MODES))
@pytest.mark.parametrize("stem", sorted(CASES))
def test_fp32_train_steps_match_reference_golden(dev, stem, mode):
    """gradients (dropout p=0) and N_STEPS fused BertAdam steps against the reference's own run."""
    from visualbert_amd.model import ModelWrapper, AttrDict
    cfg, head, sd, batch, g = load_case(stem)
    model = build_model(cfg, head, sd, dev, dtype=STRICT_MODES[mode], dropout=0.0)
    model.train()
    args = AttrDict(train_batch_size=1, learning_rate=LR, warmup_proportion=WARMUP, num_train_epochs=1,
                    gradient_accumulation_steps=1)
    mw = ModelWrapper(args, T_TOTAL, model=model)
    b = to_dev(batch, dev)
    out = mw.step(b)
    assert abs(float(out["loss"].detach()) - float(g["train_loss"])) < 1e-4
    named = dict(model.bert.named_parameters())
    gnames = set(str(x) for x in g["grad_names"])
    assert gnames <= set(named.keys())
    for n, p in named.items():
        if n not in gnames:                       # the reference leaves .grad = None (e.g. the unused pooler)
            assert float(p.grad.abs().max()) == 0.0, n
            continue
        ref_norm = float(g["grad_norm/" + n])
        gr = p.grad.detach().float().cpu()
        assert abs(float(gr.double().norm()) - ref_norm) <= 2e-3 * ref_norm + 1e-6, (n, float(gr.norm()), ref_norm)
        assert maxdiff(gr.reshape(-1)[:16], g["grad_head/" + n]) <= 2e-3 * max(ref_norm, 1e-3), n
    for _ in range(N_STEPS - 1):
        out = mw.step(b)
    assert abs(float(out["loss"].detach()) - float(g["final_loss"])) < 1e-4
    for n, p in named.items():
        assert maxdiff(p.detach().cpu().reshape(-1)[:16], g["post_head/" + n]) < 2e-6, n
        d = float((p.detach().cpu() - sd[n]).double().norm())
        if n in gnames and float(g["grad_norm/" + n]) < 1e-6:
            continue        # a gradient that is 0 in exact arithmetic (e.g. the shared bias of the 4 choices): Adam
                            # normalises its rounding noise into a +-lr step, the direction of which is noise too
        assert abs(d - float(g["delta_norm/" + n])) <= 5e-3 * float(g["delta_norm/" + n]) + 1e-7, n


@pytest.mark.parametrize("mode", ["fp32", "bf16x3", "bf16"])
def test_bert_large_width_matches_oracle(dev, mode):
    """Nothing in the path is specialised to BERT-base's 768 / 12 / 3072: one layer at BERT-large width (hidden 1024, 16 heads of 64,
    FFN 4096 -- what the reference's from_pretrained('bert-large-uncased') would build, modeling.py:44-52) against the oracle:
    logits, per-tensor gradients (relative L2) and the weights after three BertAdam steps.  Measured: fp32 3.6e-6 / 1.5e-6,
    bf16x3 2.1e-5 / 1.4e-5, bf16 1.7e-2 / 8.8e-3; asserted at about 3x (strict modes) and 2x (bf16) that."""
    from visualbert_amd.model import ModelWrapper, AttrDict
    cfg = vo.OracleConfig(vocab_size=1000, hidden_size=1024, num_hidden_layers=1, num_attention_heads=16, intermediate_size=4096,
                          visual_embedding_dim=256)
    head = "pretraining"
    sd = vo.synth_state_dict(cfg, head, 1)
    batch = vo.synth_batch(cfg, 2, 12, 4, 3, head)
    dt = {"fp32": torch.float32, "bf16x3": "bf16x3", "bf16": torch.bfloat16}[mode]
    model = build_model(cfg, head, sd, dev, dtype=dt, dropout=0.0)
    model.train()
    mw = ModelWrapper(AttrDict(train_batch_size=1, learning_rate=LR, warmup_proportion=WARMUP, num_train_epochs=1,
                               gradient_accumulation_steps=1), T_TOTAL, model=model)
    ref_sd = {k: v.clone() for k, v in sd.items()}
    state = {}
    ref_out, ref_grads = vo.train_step(ref_sd, cfg, head, batch, state, LR, WARMUP, T_TOTAL)
    out = mw.step(to_dev(batch, dev))
    lim_logit, lim_grad, lim_w = {"fp32": (1e-5, 1e-5, 2e-6), "bf16x3": (6e-5, 6e-5, 2e-6), "bf16": (3.4e-2, 1.8e-2, 1e-4)}[mode]
    assert maxdiff(out["logits"].detach().float().cpu().reshape(ref_out["logits"].shape), ref_out["logits"].detach()) <= lim_logit
    named = dict(model.bert.named_parameters())
    worst = 0.0
    for n, gr in ref_grads.items():
        rn = float(gr.double().norm())
        if rn < 1e-6:
            continue
        worst = max(worst, float((named[n].grad.detach().float().cpu() - gr).double().norm()) / rn)
    assert worst <= lim_grad, worst
    for _ in range(2):
        vo.train_step(ref_sd, cfg, head, batch, state, LR, WARMUP, T_TOTAL)
        mw.step(to_dev(batch, dev))
    for n, p_ in named.items():
        if n in ref_sd and n in ref_grads and float(ref_grads[n].double().norm()) >= 1e-6:
            assert maxdiff(p_.detach().float().cpu(), ref_sd[n]) <= lim_w, n
    if dev.type == "cuda":
        record("bert_large_width_1layer", mode, dict(grad_rel_l2_worst=worst))


# (max|dlogit| vs the bf16-emulating oracle, vs the fp32 REFERENCE golden, |dloss|): 1.5 x the values measured on MI355X
# (profiles/r02_parity_small.json)
#   measured: micro_pretraining 5.6e-3 / 3.2e-3 / 5.5e-4   tiny_pretraining 7.2e-3 / 6.4e-3 / 5.6e-4
#             micro_bypass      5.2e-3 / 3.8e-3 / 1.2e-3   micro_align      5.8e-3 / 5.5e-3 / 6e-5   (|dloss|: 3 x, few samples)
BF16_FWD_BOUNDS = {"micro_pretraining": (8.5e-3, 4.8e-3, 1.7e-3), "tiny_pretraining": (1.1e-2, 9.7e-3, 1.7e-3),
                   "micro_bypass": (7.9e-3, 5.7e-3, 3.6e-3), "micro_align": (8.8e-3, 8.2e-3, 1.0e-3)}


@pytest.mark.parametrize("stem", ["micro_pretraining", "tiny_pretraining", "micro_bypass", "micro_align"])
def test_bf16_matches_bf16_oracle(dev, stem):
    """bf16 kernels against the oracle with bf16 rounding at the same storage points (DESIGN.md numeric
    contract); the bf16-vs-fp32-reference gap is printed, not asserted."""
    cfg, head, sd, batch, g = load_case(stem)
    model = build_model(cfg, head, sd, dev, dtype=torch.bfloat16)
    model.eval()
    with torch.no_grad():
        out = model(**to_dev(batch, dev))
        ref = vo.objective_forward(sd, cfg, head, mode="bf16", **batch)
    lg = out["logits"].float().cpu()
    err = float((lg - ref["logits"]).abs().max())
    gap = maxdiff(lg[:, :, ::LOGIT_STRIDE], g["logits_strided"])
    print("bf16 logits: vs bf16-oracle %.3e, vs fp32 reference %.3e (absmax %.2f)" % (err, gap, float(g["logits_absmax"])))
    dl = abs(float(out["loss"].detach()) - float(ref["loss"]))
    record("bf16_small_forward", stem, dict(vs_bf16_oracle=err, vs_fp32_reference=gap, dloss_vs_bf16_oracle=dl))
    lim = BF16_FWD_BOUNDS[stem]
    assert err < lim[0], err
    assert gap < lim[1], gap
    assert dl < lim[2], dl


def test_layer_call_plans_follow_the_parameters(dev):
    """ops._LayerPlan caches a layer's pointer arrays (round 6: the small-batch step had become host-bound).  The cache must notice every
    way the parameters can change under it: a write through torch (version counter), a state-dict load, an optimizer step through the raw
    kernel (shadows refreshed in place: nothing to notice, same addresses), a new arena (.to(device) / a rebuilt model) -- each time the
    next forward must equal a FRESH model built from the same weights, bit for bit."""
    from visualbert_amd import ops
    cfg, head, sd, batch, g = load_case("micro_pretraining")
    b = to_dev(batch, dev)

    def fresh_logits(state):
        m = build_model(cfg, head, state, dev, dtype=torch.bfloat16, dropout=0.0)
        m.eval()
        with torch.no_grad():
            return m(**b)["logits"].float().cpu()

    model = build_model(cfg, head, sd, dev, dtype=torch.bfloat16, dropout=0.0)
    model.eval()
    with torch.no_grad():
        first = model(**b)["logits"].float().cpu()
    layer = model.bert.bert.encoder.layer[0]
    assert layer.__dict__.get("_vb_plan") is not None, "arena-managed bf16 layer: the plan is expected to exist"
    assert torch.equal(first, fresh_logits(sd))
    # (1) a write through torch: one FFN weight and one LayerNorm bias of layer 0, the packed query weight of layer 1
    sd2 = {k: v.clone() for k, v in sd.items()}
    with torch.no_grad():
        layer.output.dense.weight.mul_(1.5)
        layer.attention.output.LayerNorm.bias.add_(0.25)
        model.bert.bert.encoder.layer[1].attention.self.query.weight.mul_(0.5)
    sd2["bert.encoder.layer.0.output.dense.weight"] *= 1.5
    sd2["bert.encoder.layer.0.attention.output.LayerNorm.bias"] += 0.25
    sd2["bert.encoder.layer.1.attention.self.query.weight"] *= 0.5
    with torch.no_grad():
        second = model(**b)["logits"].float().cpu()
    assert not torch.equal(second, first)
    assert torch.equal(second, fresh_logits(sd2))
    # (2) a state-dict load back to the original weights
    own = model.bert.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            own[k].copy_(v)
    with torch.no_grad():
        third = model(**b)["logits"].float().cpu()
    assert torch.equal(third, first)
    # (3) training steps through the fused optimizer (raw writes into the arena + shadows refreshed by the kernel), then eval again:
    #     equal to a fresh model loaded from the trained model's state dict
    model.train()
    from visualbert_amd.model import AttrDict, ModelWrapper
    mw = ModelWrapper(AttrDict(train_batch_size=b["bert_input_ids"].size(0), learning_rate=1e-3, warmup_proportion=0.1, num_train_epochs=1,
                               gradient_accumulation_steps=1), 100 * b["bert_input_ids"].size(0), model=model)
    for _ in range(2):
        mw.step(b)
    model.eval()
    with torch.no_grad():
        fourth = model(**b)["logits"].float().cpu()
    trained = {k: v.detach().float().cpu().clone() for k, v in model.bert.state_dict().items() if k in sd}
    assert not torch.equal(fourth, first)
    assert torch.equal(fourth, fresh_logits(trained))


def test_dropout_training_step_runs_and_is_seed_deterministic(dev):
    cfg, head, sd, batch, g = load_case("micro_pretraining")
    losses = []
    for rep in range(2):
        torch.manual_seed(123)
        from visualbert_amd import ops
        ops.reset_seed_counter()
        model = build_model(cfg, head, sd, dev)
        model.train()
        out = model(**to_dev(batch, dev))
        out["loss"].backward()
        losses.append(float(out["loss"].detach()))
        gn = float(model.bert.arena.grad.norm())
        assert np.isfinite(gn) and gn > 0
    assert abs(losses[0] - losses[1]) < 1e-5     # same masks; only atomic summation order may differ
    assert abs(losses[0] - float(g["loss"])) > 1e-6      # dropout really changed the forward


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_weights_match_reference_golden(dev, dtype):
    """output_attention_weights=True (modeling.py:1428-1442): the forward returns the per-layer attention probabilities
    [B, nh, S, S] and loss None; fp32 kernels against the real reference's tensors, bf16 at bf16 tolerance."""
    import os
    from golden_util import GOLDEN_DIR
    from visualbert_amd.modeling import BertConfig
    from visualbert_amd.model import VisualBERTFixedImageEmbedding
    g = np.load(os.path.join(GOLDEN_DIR, "micro_attention_weights.npz"))
    B, T, R, seed = [int(x) for x in g["meta"]]
    cfg = vo.OracleConfig(**vo.CONFIGS["micro"])
    sd = vo.synth_state_dict(cfg, "pretraining", seed)
    batch = vo.synth_batch(cfg, B, T, R, seed, "pretraining")
    bc = BertConfig(cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                    num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                    max_position_embeddings=cfg.max_position_embeddings, type_vocab_size=cfg.type_vocab_size)
    model = VisualBERTFixedImageEmbedding(config=bc, training_head_type="pretraining", compute_dtype=dtype,
                                          visual_embedding_dim=cfg.visual_embedding_dim,
                                          output_attention_weights=True).to(dev)
    own = model.bert.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            own[k].copy_(v)
    model.eval()
    with torch.no_grad():
        out = model(**to_dev(batch, dev))
    assert out["loss"] is None
    assert len(out["attention_weights"]) == cfg.num_hidden_layers
    for i, w in enumerate(out["attention_weights"]):
        ref = g["attention_weights/%d" % i]
        assert tuple(w.shape) == ref.shape
        assert maxdiff(w.cpu(), ref) < (1e-5 if dtype == torch.float32 else 2e-2)
        assert float((w.sum(-1) - 1).abs().max()) < 1e-5


@pytest.mark.parametrize("stem", ["micro_multichoice", "micro_flickr"])
def test_bf16_small_heads_match_bf16_oracle(dev, stem):
    """the N4 heads with bf16 kernels: loss against the bf16-emulating oracle; the flickr counts stay exact unless two
    regions score within bf16 noise of each other (not the case for the fixture)."""
    cfg, head, sd, batch, g = load_case(stem)
    model = build_model(cfg, head, sd, dev, dtype=torch.bfloat16)
    model.eval()
    with torch.no_grad():
        out = model(**to_dev(batch, dev))
        ref = vo.objective_forward(sd, cfg, head, mode="bf16", **batch)
    assert abs(float(out["loss"].detach()) - float(ref["loss"])) < 2e-2
    if head == "flickr":
        assert float(out["entity_num"]) == float(ref["entity_num"])
        assert abs(float(out["upperbound_accuracy"]) - float(ref["upperbound_accuracy"])) < 1e-6
    else:
        assert maxdiff(out["logits"].float().cpu(), ref["logits"]) < 2e-2


# per-tensor relative L2 (median, worst) of the bf16 gradients against the bf16-emulating oracle: 1.5 x measured
#   measured (median / worst): micro_pretraining 7.8e-3 / 1.06e-2, tiny_pretraining 7.2e-3 / 7.5e-2 (pooler bias), micro_bypass
#   6.9e-3 / 1.38e-2, micro_align 6.6e-3 / 8.2e-3, micro_flickr 1.66e-2 / 7.3e-2 (projection bias)
BF16_GRAD_BOUNDS = {"micro_pretraining": (0.0117, 0.0159), "tiny_pretraining": (0.0109, 0.113), "micro_bypass": (0.0103, 0.0208),
                    "micro_align": (0.0099, 0.0123), "micro_flickr": (0.0249, 0.109)}


@pytest.mark.parametrize("stem", ["micro_pretraining", "tiny_pretraining", "micro_bypass", "micro_align", "micro_flickr"])
def test_bf16_gradients_track_bf16_oracle(dev, stem):
    """bf16 backward (W^T shadows + LDS-direct dgrad, split-K wgrad, fused layer call): every parameter's gradient against
    the bf16-emulating oracle's, as a relative L2 error per tensor (a wrong scale of 1.04 on any tensor fails it);
    bounds = 1.5 x measured (profiles/r02_parity_small.json)."""
    cfg, head, sd, batch, g = load_case(stem)
    model = build_model(cfg, head, sd, dev, dtype=torch.bfloat16, dropout=0.0)
    model.train()
    out = model(**to_dev(batch, dev))
    out["loss"].backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = vo.objective_forward(leaves, cfg, head, mode="bf16", **batch)
    ref["loss"].backward()
    rels = {}
    for n, p in model.bert.named_parameters():
        rg = leaves[n].grad
        if rg is None:
            continue
        mg = p.grad.detach().float().cpu()
        rn = float(rg.norm())
        if rn < 1e-7 or n.endswith("key.bias"):     # d/d(key bias) is identically 0 (softmax shift invariance): noise
            continue
        rels[n] = float((mg - rg).norm()) / rn
    order = sorted(rels, key=rels.get)
    rec = dict(worst=rels[order[-1]], worst_name=order[-1], median=rels[order[len(order) // 2]])
    record("bf16_small_grads", stem, rec)
    print("bf16 gradients vs bf16 oracle: relative L2 median %.4f, worst %.4f (%s)" % (rec["median"], rec["worst"], rec["worst_name"]))
    lim = BF16_GRAD_BOUNDS[stem]
    assert rec["median"] <= lim[0], rec
    assert rec["worst"] <= lim[1], rec


def test_bert_base_config2_logits_vs_oracle(dev):
    """BASELINE.json configs[1] at its real size: BERT-base 12L/768, 36 regions x 2048-d + 128 tokens (S=164),
    ragged masks, B=2.  fp32 kernels must reproduce the oracle's logits within the north-star's 1e-3
    (the oracle itself is pinned to the real reference by tests/golden); the bf16 kernels' gap to the fp32
    reference is measured and printed (~2e-2, not hidden)."""
    if dev.type != "cuda":
        pytest.skip("full-size case: GPU only (the simulator would take hours)")
    cfg = vo.OracleConfig(**vo.CONFIGS["base"])
    head = "pretraining"
    sd = vo.synth_state_dict(cfg, head, 11)
    batch = vo.synth_batch(cfg, 2, 128, 36, 11, head, ragged=True)
    with torch.no_grad():
        ref = vo.objective_forward(sd, cfg, head, mode="fp32", **batch)
    model = build_model(cfg, head, sd, dev)
    model.eval()
    with torch.no_grad():
        out = model(**to_dev(batch, dev))
    lg = out["logits"].float().cpu()
    err = float((lg - ref["logits"]).abs().max())
    print("BERT-base fp32: max|dlogit| %.3e (absmax %.2f), |dloss| %.3e" %
          (err, float(ref["logits"].abs().max()), abs(float(out["loss"].detach()) - float(ref["loss"]))))
    assert err < 1e-3, err
    assert abs(float(out["loss"].detach()) - float(ref["loss"])) < 1e-4
    assert torch.equal(lg.argmax(-1), ref["logits"].argmax(-1))          # token indexing bit-exact
    assert torch.equal(out["seq_relationship_score"].argmax(-1).cpu(), ref["seq_relationship_score"].argmax(-1))
    model.bert.set_compute_dtype(torch.bfloat16)
    with torch.no_grad():
        out16 = model(**to_dev(batch, dev))
        ref16 = vo.objective_forward(sd, cfg, head, mode="bf16", **batch)
    lg16 = out16["logits"].float().cpu()
    gap = float((lg16 - ref["logits"]).abs().max())
    err16 = float((lg16 - ref16["logits"]).abs().max())
    print("BERT-base bf16: max|dlogit| vs fp32 reference %.3e, vs bf16-emulating oracle %.3e" % (gap, err16))
    record("bf16_base_b2", "logits", dict(vs_fp32_reference=gap, vs_bf16_oracle=err16))
    # measured on MI355X: 3.7e-2 vs the fp32 reference, 3.3e-2 vs the bf16-emulating oracle (1.5 x asserted); the B = 16
    # case with mean / p99.9 / top-1 statistics is tests/test_parity_at_scale.py
    assert gap < 5.5e-2 and err16 < 4.9e-2


def test_nlvr2_reference_shape_long_sequence(dev):
    """NLVR2 as the reference really feeds it (SURVEY 8d, config 5 note): two images x 144 regions of 1024-d features +
    128 tokens, S = 416 -- past the S = 164 the kernels are tuned for.  bf16 training step against the bf16-emulating
    oracle: loss, logits, and the direction of two gradients at opposite ends of the network."""
    if dev.type != "cuda":
        pytest.skip("BERT-base at S=416: GPU only")
    kw = dict(vo.CONFIGS["base"], visual_embedding_dim=1024, num_hidden_layers=4)
    cfg = vo.OracleConfig(**kw)
    head = "nlvr"
    sd = vo.synth_state_dict(cfg, head, 21)
    batch = vo.synth_batch(cfg, 2, 128, 288, 21, head, ragged=True)
    model = build_model(cfg, head, sd, dev, dtype=torch.bfloat16, dropout=0.0)
    model.train()
    out = model(**to_dev(batch, dev))
    out["loss"].backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = vo.objective_forward(leaves, cfg, head, mode="bf16", **batch)
    ref["loss"].backward()
    assert abs(float(out["loss"].detach()) - float(ref["loss"].detach())) < 2e-2
    assert maxdiff(out["logits"].detach().float().cpu(), ref["logits"].detach()) < 3e-2
    named = dict(model.bert.named_parameters())
    for n in ("classifier.weight", "bert.encoder.layer.0.attention.self.query.weight",
              "bert.embeddings.projection.weight"):
        rg, mg = leaves[n].grad, named[n].grad.detach().float().cpu()
        cos = float((rg * mg).sum() / (rg.norm() * mg.norm() + 1e-30))
        assert cos > 0.98, (n, cos)


@pytest.mark.parametrize("mode", sorted(STRICT_MODES))
def test_nlvr2_reference_shape_strict_modes(dev, mode):
    """the same S = 416 step in the two modes that must meet the north-star's tolerance: fp32 kernels and bf16x3, against the
    fp32 oracle (the reference's arithmetic) -- the key-tiled attention forward / dQ pass (csrc/attention.hip) is what lets the
    strict gate cover the reference's real long-sequence shape (modeling.py:83 max_position_embeddings = 512,
    configs/nlvr2/fine-tune.json: 2 x 144 regions + 128 tokens)."""
    if dev.type != "cuda":
        pytest.skip("BERT-base width at S=416: GPU only")
    kw = dict(vo.CONFIGS["base"], visual_embedding_dim=1024, num_hidden_layers=4)
    cfg = vo.OracleConfig(**kw)
    head = "nlvr"
    sd = vo.synth_state_dict(cfg, head, 21)
    batch = vo.synth_batch(cfg, 2, 128, 288, 21, head, ragged=True)
    model = build_model(cfg, head, sd, dev, dtype=STRICT_MODES[mode], dropout=0.0)
    model.train()
    out = model(**to_dev(batch, dev))
    out["loss"].backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = vo.objective_forward(leaves, cfg, head, mode="fp32", **batch)
    ref["loss"].backward()
    assert abs(float(out["loss"].detach()) - float(ref["loss"].detach())) < 1e-4
    assert maxdiff(out["logits"].detach().float().cpu(), ref["logits"].detach()) < 1e-3       # north_star; measured ~1e-5
    assert maxdiff(out["logits"].detach().float().cpu(), ref["logits"].detach()) < 1e-4
    named = dict(model.bert.named_parameters())
    for n in ("classifier.weight", "bert.encoder.layer.0.attention.self.query.weight",
              "bert.encoder.layer.3.output.dense.weight", "bert.embeddings.projection.weight"):
        rg, mg = leaves[n].grad, named[n].grad.detach().float().cpu()
        assert float((rg - mg).norm()) <= 2e-3 * float(rg.norm()), (n, float((rg - mg).norm()) / float(rg.norm()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_sparse_mlm_head_matches_dense_head(dev, dtype):
    """the MLM head over the labelled positions only gives the dense head's loss, the dense
    head's logits at those positions and the dense head's parameter gradients."""
    cfg, head, sd, batch, g = load_case("micro_pretraining")
    outs = []
    for sparse in (False, True):
        model = build_model(cfg, head, sd, dev, dtype=dtype, dropout=0.0)
        model.train()
        model.bert.sparse_mlm_head = sparse
        model.bert.arena.zero_grad()
        out = model(**to_dev(batch, dev))
        out["loss"].backward()
        outs.append((out, model.bert.arena.grad.detach().float().cpu().clone()))
    (od, gd), (os_, gs) = outs
    assert abs(float(od["loss"].detach()) - float(os_["loss"].detach())) <= (1e-5 if dtype == torch.float32 else 2e-3)
    rows = os_["logits_rows"].cpu()
    V = od["logits"].size(-1)
    dense_rows = od["logits"].reshape(-1, V).float().cpu()[rows]
    lerr = (os_["logits"].float().cpu() - dense_rows).abs().max().item()
    assert lerr <= (1e-4 if dtype == torch.float32 else 5e-2), lerr
    gerr = (gd - gs).abs().max().item()
    assert gerr <= (2e-5 if dtype == torch.float32 else 2e-2) * max(1.0, gd.abs().max().item()), gerr


@pytest.mark.parametrize("name,kw", [("WarmupCosineSchedule", dict(warmup=0.1, cycles=0.5)),
                                     ("WarmupCosineWithWarmupRestartsSchedule", dict(warmup=0.05, cycles=4.0))])
def test_bert_adam_host_evaluated_schedules(dev, name, kw):
    """schedules without a device formula: the fused BertAdam takes the step's multiplier from the host.  Four steps
    against the oracle's BertAdam restatement driven by the reference's own multipliers (tests/golden/schedules.json)."""
    import json
    import os
    from visualbert_amd import optimization as opt
    from golden_util import GOLDEN_DIR
    fx = json.load(open(os.path.join(GOLDEN_DIR, "schedules.json")))
    ref_lr = [c for c in fx["cases"] if c["schedule"] == name][0]["lr"]
    cfg, head, sd, batch, g = load_case("micro_nlvr")
    model = build_model(cfg, head, sd, dev, dropout=0.0)
    model.train()
    named = [n for n in model.named_parameters() if "pooler" not in n[0]]
    no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
    groups = [{"params": [p for n, p in named if not any(nd in n for nd in no_decay)], "weight_decay": 0.01},
              {"params": [p for n, p in named if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
    sch = getattr(opt, name)(t_total=fx["t_total"], **kw)
    optim = opt.BertAdam(groups, lr=1e-3, schedule=sch)
    ref_sd = copy.deepcopy(sd)
    state = {}
    b = to_dev(batch, dev)
    for step in range(4):
        optim.zero_grad()
        out = model(**b)
        out["loss"].backward()
        optim.step()
        leaves = {k: v.detach().clone().requires_grad_(True) for k, v in ref_sd.items()}
        ro = vo.objective_forward(leaves, cfg, head, mode="fp32", **batch)
        ro["loss"].backward()
        grads = {k: v.grad for k, v in leaves.items() if v.grad is not None}
        with torch.no_grad():
            vo.bert_adam_step(ref_sd, grads, state, 1e-3, 0.0, -1, schedule=lambda s: ref_lr[s])
        assert abs(float(out["loss"].detach()) - float(ro["loss"].detach())) < 1e-4, step
    for n, p in model.bert.named_parameters():
        assert maxdiff(p.detach().cpu(), ref_sd[n]) < 5e-6, n
    assert optim.state_dict()["state"][0]["step"] == 4


def test_checkpoint_round_trip_resumes_identically(dev, tmp_path):
    """ModelWrapper.save_checkpoint / restore_checkpoint (models/model_wrapper.py:163-199 file names and keys): a fresh
    model + optimizer restored from disk continues exactly like the one that kept running (weights, Adam moments,
    per-tensor step counters, LR schedule position)."""
    from visualbert_amd.model import ModelWrapper, AttrDict
    cfg, head, sd, batch, g = load_case("micro_pretraining")
    args = AttrDict(train_batch_size=1, learning_rate=LR, warmup_proportion=WARMUP, num_train_epochs=1,
                    gradient_accumulation_steps=1)
    b = to_dev(batch, dev)
    mw = ModelWrapper(args, T_TOTAL, model=build_model(cfg, head, sd, dev, dropout=0.0))
    mw.train()
    for _ in range(2):
        mw.step(b)
    mw.save_checkpoint(str(tmp_path), 3, [0.5, 0.6])
    mw.save_checkpoint_step(str(tmp_path), 7, 1)
    mw2 = ModelWrapper(args, T_TOTAL, model=build_model(cfg, head, vo.synth_state_dict(cfg, head, 99), dev, dropout=0.0))
    mw2.train()
    epoch, metrics = mw2.restore_checkpoint(str(tmp_path))
    assert epoch == 4 and metrics == [0.5, 0.6]
    la, lb = float(mw.step(b)["loss"].detach()), float(mw2.step(b)["loss"].detach())
    assert abs(la - lb) < 1e-6
    for (n, p), (_, q) in zip(mw.model.bert.named_parameters(), mw2.model.bert.named_parameters()):
        assert maxdiff(p.detach().cpu(), q.detach().cpu()) < 1e-7, n
    assert mw2.optimizer.state_dict()["state"][0]["step"] == 3
    fresh = ModelWrapper(args, T_TOTAL, model=build_model(cfg, head, sd, dev, dropout=0.0))
    assert fresh.restore_checkpoint(str(tmp_path / "nothing_here")) == (0, [])


def test_collated_pinned_batch_streams_and_trains(dev):
    """host data path end to end: collate_pretraining builds the padded batch in pinned memory,
    FeatureStager streams it to HBM on a side stream, the model takes the kwargs as they are."""
    if dev.type != "cuda":
        pytest.skip("pinned memory + async copies: GPU only")
    from visualbert_amd.data import collate_pretraining, FeatureStager
    cfg, head, sd, _, _ = load_case("micro_pretraining")
    g = torch.Generator().manual_seed(2)
    B = 6
    ids_a = [torch.randint(5, cfg.vocab_size, (int(n),), generator=g) for n in torch.randint(2, 9, (B,), generator=g)]
    ids_b = [torch.randint(5, cfg.vocab_size, (int(n),), generator=g) for n in torch.randint(1, 7, (B,), generator=g)]
    feats = [torch.rand(int(n), cfg.visual_embedding_dim, generator=g) for n in torch.randint(2, 6, (B,), generator=g)]
    host = collate_pretraining(ids_a, ids_b, [True, False] * 3, feats, cfg.vocab_size, 3, 1, 2, generator=g)
    assert all(v.is_pinned() for v in host.values())
    batch, ev = FeatureStager(dev).stage(host)
    torch.cuda.current_stream().wait_event(ev)
    model = build_model(cfg, head, sd, dev, dropout=0.0)
    model.train()
    out = model(**batch)
    out["loss"].backward()
    ref = vo.objective_forward(sd, cfg, head, mode="fp32", **{k: v.clone() for k, v in host.items()})
    assert abs(float(out["loss"].detach()) - float(ref["loss"])) < 1e-4


def test_captions_and_region_files_to_training_steps(dev, tmp_path):
    """the whole ingest path in front of the hot path, as one pipeline (SURVEY 8f N2 + N3): caption TEXT -> WordPieceEncoder ->
    collate_pretraining (masking, [CLS]/[SEP], padding) -> per-image region .npy files -> RegionFeatureStore (pinned slab) ->
    FeatureStager (side-stream H2D) -> ModelWrapper.step.  The first loss equals the oracle's on the same host batch; a few
    steps on the same batch bring it down."""
    if dev.type != "cuda":
        pytest.skip("pinned memory + async copies: GPU only")
    import numpy as np
    from visualbert_amd.data import collate_pretraining, FeatureStager, RegionFeatureStore
    from visualbert_amd.model import ModelWrapper, AttrDict
    from visualbert_amd.tokenization import WordPieceEncoder
    cfg, head, sd, _, _ = load_case("micro_pretraining")
    specials = ["[PAD]", "[CLS]", "[SEP]", "[MASK]", "[UNK]"]
    words = ["a", "the", "man", "woman", "dog", "ride", "riding", "horse", "on", "in", "park", "play", "##s", "##ing", "##ed", ".", ","]
    vocab = specials + words + ["w%d" % i for i in range(cfg.vocab_size - len(specials) - len(words))]
    assert len(vocab) == cfg.vocab_size
    enc = WordPieceEncoder({t: i for i, t in enumerate(vocab)})
    captions = [("A man riding a horse.", "the dog plays in the park"), ("The woman rides.", None),
                ("a dog, a horse, a man", "riding on the horse"), ("the park", "a man played")]
    ids_a = [torch.tensor(enc.encode(a), dtype=torch.int64) for a, _ in captions]
    ids_b = [torch.tensor(enc.encode(b), dtype=torch.int64) if b else None for _, b in captions]
    assert enc.tokenize("plays riding") == ["play", "##s", "riding"]
    rng = np.random.RandomState(3)
    image_ids = [11, 222, 3333, 44]
    for i, r in zip(image_ids, (4, 2, 5, 3)):
        np.save(str(tmp_path / ("COCO_train2014_%012d.npy" % i)), rng.rand(r, cfg.visual_embedding_dim).astype(np.float32))
    store = RegionFeatureStore(str(tmp_path), "train")
    region = store.read_batch(image_ids)
    feats = [region["image_feat_variable"][b, :int(region["image_dim_variable"][b])] for b in range(len(image_ids))]
    g = torch.Generator().manual_seed(7)
    host = collate_pretraining(ids_a, ids_b, [True, False, True, False], feats, cfg.vocab_size, vocab.index("[MASK]"),
                               vocab.index("[CLS]"), vocab.index("[SEP]"), probability=0.5, generator=g)
    assert all(v.is_pinned() for v in host.values())
    assert (host["masked_lm_labels"] >= 0).any()
    stager = FeatureStager(dev)
    batch, ev = stager.stage(host)
    torch.cuda.current_stream().wait_event(ev)
    model = build_model(cfg, head, sd, dev, dropout=0.0)
    mw = ModelWrapper(AttrDict(train_batch_size=4, learning_rate=2e-3, warmup_proportion=0.1, num_train_epochs=1), 40,
                      model=model)
    model.train()
    losses = [float(mw.step(batch)["loss"].detach()) for _ in range(6)]
    ref = vo.objective_forward(sd, cfg, head, mode="fp32", **{k: v.clone() for k, v in host.items()})
    assert abs(losses[0] - float(ref["loss"])) < 1e-4
    assert losses[-1] < losses[0] - 0.05, losses


def test_degenerate_batches_match_oracle(dev):
    """edges of the input contract: a sample whose regions are ALL padding (image_dim = 0: its visual slots are masked
    keys but still produce rows), a sample with a single real token, and a batch without any labelled token -- the
    reference's CrossEntropyLoss(ignore_index=-1) then averages over nothing and returns NaN (modeling.py:1471-1473)."""
    cfg, head, sd, batch, g = load_case("micro_pretraining")
    model = build_model(cfg, head, sd, dev)
    model.eval()
    b = {k: v.clone() for k, v in batch.items()}
    b["image_dim_variable"][0] = 0                                     # no valid region at all
    b["image_feat_variable"][0] = 0
    b["bert_input_mask"][1] = 0
    b["bert_input_mask"][1, 0] = 1                                     # one real token
    b["bert_input_ids"][1, 1:] = 0
    b["masked_lm_labels"][1] = -1
    b["masked_lm_labels"][1, 0] = int(b["bert_input_ids"][1, 0])
    with torch.no_grad():
        out = model(**to_dev(b, dev))
        ref = vo.objective_forward(sd, cfg, head, mode="fp32", **b)
    assert maxdiff(out["logits"].float().cpu(), ref["logits"]) < 1e-4
    assert abs(float(out["loss"]) - float(ref["loss"])) < 1e-4
    assert torch.equal(out["logits"].float().cpu().argmax(-1), ref["logits"].argmax(-1))
    b2 = {k: v.clone() for k, v in batch.items()}
    b2["masked_lm_labels"][:] = -1                                     # nothing to predict
    with torch.no_grad():
        out2 = model(**to_dev(b2, dev))
        ref2 = vo.objective_forward(sd, cfg, head, mode="fp32", **b2)
    assert bool(torch.isnan(ref2["masked_lm_loss"])) and bool(torch.isnan(out2["masked_lm_loss"].cpu()))
    assert maxdiff(out2["seq_relationship_score"].cpu(), ref2["seq_relationship_score"]) < 1e-4


def test_text_only_batch_matches_oracle(dev):
    """no image features at all (image_feat_variable = None): plain BERT over the text (models/model.py:269-270,
    modeling.py:1213-1221, 1427-1428) -- forward, loss and gradients against the oracle."""
    cfg, head, sd, batch, g = load_case("micro_pretraining")
    b = {k: v for k, v in batch.items() if not k.startswith("image_")}
    model = build_model(cfg, head, sd, dev, dropout=0.0)
    model.train()
    out = model(**to_dev(b, dev))
    out["loss"].backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = vo.objective_forward(leaves, cfg, head, mode="fp32", **b)
    ref["loss"].backward()
    assert out["logits"].shape == ref["logits"].shape
    assert maxdiff(out["logits"].detach().float().cpu(), ref["logits"].detach()) < 1e-4
    assert abs(float(out["loss"].detach()) - float(ref["loss"].detach())) < 1e-4
    named = dict(model.bert.named_parameters())
    for n in ("bert.embeddings.word_embeddings.weight", "bert.encoder.layer.0.attention.self.query.weight",
              "cls.predictions.transform.dense.weight"):
        rg = leaves[n].grad
        assert maxdiff(named[n].grad.detach().cpu(), rg) <= 2e-3 * max(float(rg.norm()), 1e-3), n
    assert float(named["bert.embeddings.projection.weight"].grad.abs().max()) == 0.0   # untouched without regions


def test_adam_keeps_stepping_a_tensor_once_it_has_had_a_gradient(dev):
    """optimization.py:254-255 skips a parameter only while `p.grad is None`.  After a parameter's first backward the
    reference's zero_grad() leaves a ZERO tensor behind, so on a later text-only batch the visual tables and the region
    projection still take a step: moments decay and -- for decayed tensors -- the weight decay is applied.  A tensor no
    batch ever touched (here: nothing, all are touched by the first batch; see test_text_only_batch_matches_oracle for
    the never-touched side) stays skipped.  Sequence: one batch with regions, then two text-only batches, against the
    oracle's BertAdam driven with zero gradients in place of None for every tensor that has had a gradient."""
    from visualbert_amd.model import ModelWrapper, AttrDict
    cfg, head, sd, batch, g = load_case("micro_pretraining")
    text_only = {k: v for k, v in batch.items() if not k.startswith("image_")}
    model = build_model(cfg, head, sd, dev, dropout=0.0)
    mw = ModelWrapper(AttrDict(train_batch_size=1, learning_rate=1e-3, warmup_proportion=0.1, num_train_epochs=1,
                               gradient_accumulation_steps=1), 100, model=model)
    model.train()
    ref_sd, state, seen = copy.deepcopy(sd), {}, set()
    for b in (batch, text_only, text_only):
        mw.step(to_dev(b, dev))
        leaves = {k: v.detach().clone().requires_grad_(True) for k, v in ref_sd.items()}
        vo.objective_forward(leaves, cfg, head, mode="fp32", **b)["loss"].backward()
        grads = {}
        for k, v in leaves.items():
            if v.grad is not None:
                seen.add(k)
                grads[k] = v.grad
            elif k in seen:
                grads[k] = torch.zeros_like(v)              # what .grad holds upstream after zero_grad()
        with torch.no_grad():
            vo.bert_adam_step(ref_sd, grads, state, 1e-3, 0.1, 100)
    named = dict(model.bert.named_parameters())
    moved = float((sd["bert.embeddings.projection.weight"] - ref_sd["bert.embeddings.projection.weight"]).abs().max())
    assert moved > 0
    for n in ("bert.embeddings.projection.weight", "bert.embeddings.token_type_embeddings_visual.weight",
              "bert.embeddings.position_embeddings_visual.weight", "bert.embeddings.word_embeddings.weight"):
        assert maxdiff(named[n].detach().cpu(), ref_sd[n]) <= 2e-6, n
