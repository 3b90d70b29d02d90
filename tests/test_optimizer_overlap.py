"""BertAdam following the backward pass bucket by bucket (visualbert_amd.optimization.BertAdam.overlap_with_backward).

The reference steps every tensor on its own (per-tensor clip, per-tensor counter: pytorch_pretrained_bert/optimization.py:254-304),
so stepping an encoder layer's tensors as soon as their gradients are final -- on a second stream, while the layers below are
still in their backward pass -- must leave the SAME BITS as the one-pass step: weights, both moments, step counters, the bf16
shadows the GEMMs read and their transposed copies.
"""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _wrapper(dev, overlap, gas=1, dtype=torch.bfloat16, layers=3, zero=True):
    from oracle import visualbert_oracle as vo
    from visualbert_amd.modeling import BertConfig
    from visualbert_amd.model import VisualBERTFixedImageEmbedding, ModelWrapper, AttrDict
    cfg = vo.OracleConfig(**dict(vo.CONFIGS["micro"], num_hidden_layers=layers))
    bc = BertConfig(cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                    num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size,
                    hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = VisualBERTFixedImageEmbedding(config=bc, training_head_type="pretraining",
                                          visual_embedding_dim=cfg.visual_embedding_dim, compute_dtype=dtype).to(dev)
    sd = vo.synth_state_dict(cfg, "pretraining", 11)
    own = model.bert.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            own[k].copy_(v)
    model.train()
    mw = ModelWrapper(AttrDict(train_batch_size=4, learning_rate=2e-3, warmup_proportion=0.1, num_train_epochs=1,
                               gradient_accumulation_steps=gas, overlap_optimizer=overlap), 400, model=model)
    mw._after_weights_changed()
    if overlap:
        assert mw.optimizer._ov["zero"] is False            # ModelWrapper leaves the gradients readable after step(), like the reference
        mw.optimizer._ov["zero"] = zero
    batches = [{k: v.to(dev) for k, v in vo.synth_batch(cfg, 4, 12, 5, 20 + i, "pretraining").items()} for i in range(4)]
    return mw, batches


def _state(mw):
    f = mw.optimizer.fused()
    a = f["arena"]
    return dict(data=a.data.clone(), m=f["m"].clone(), v=f["v"].clone(), steps=f["steps"].clone(), shadow=a.shadow.clone(),
                shadow_t=a.shadow_t.clone(), grad=a.grad.clone())


@pytest.mark.gpu
@pytest.mark.parametrize("zero", [True, False])
def test_overlapped_step_leaves_the_bits_of_the_one_pass_step(dev, zero):
    """the same gradients into both optimizers (a backward pass is not bit-reproducible from run to run: the embedding tables'
    gradients are summed with atomics), the hooks fired by hand in the order backward fires them."""
    plain, _ = _wrapper(dev, False)
    over, _ = _wrapper(dev, True)
    over.optimizer._ov["zero"] = zero
    ap, ao = plain.model.bert.arena, over.model.bert.arena
    assert torch.equal(ap.data, ao.data)
    gen = torch.Generator().manual_seed(5)
    real = torch.zeros(ap.grad.numel())                     # the arena pads every tensor to 64 elements: no gradient there
    for p, o in zip(ap.params, ap.offsets):
        real[o:o + p.numel()] = 1.0
    for step in range(3):
        g = (torch.randn(ap.grad.numel(), generator=gen) * real * (10.0 ** (step - 2))).to(dev)      # small, medium, clipped
        for w in (plain, over):
            a = w.model.bert.arena
            a.zero_grad()
            a.grad.copy_(g)
            a.touched.update(id(p) for p in a.params)
        plain.optimizer.step()
        over.optimizer.arm_overlap()
        for i in (2, 1, 0):
            over.optimizer._layer_grads_final(i)
        assert over.optimizer._ov["done"] == ["layer2", "layer1", "layer0"]
        over.optimizer.step()
        sp, so = _state(plain), _state(over)
        for k in ("data", "m", "v", "steps", "shadow", "shadow_t"):
            assert torch.equal(sp[k], so[k]), (step, k)
        assert float(sp["grad"].abs().max()) > 0.0
        assert (float(so["grad"].abs().max()) == 0.0) == zero
        assert (ao._clean_token is not None) == zero
    assert int(so["steps"].max()) == 3 and float((so["data"] - g.new_zeros(())).abs().max()) > 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("gas", [1, 2])
def test_training_with_the_overlapped_step_follows_the_one_pass_trajectory(dev, gas):
    """the real step (ModelWrapper.step) with and without the overlap: same trajectory up to the run-to-run noise of the
    atomically summed embedding gradients; gradient-accumulation micro-steps move nothing."""
    plain, batches = _wrapper(dev, False, gas)
    over, _ = _wrapper(dev, True, gas)
    assert over.optimizer._ov is not None and plain.optimizer._ov is None
    a = over.model.bert.arena
    for i, b in enumerate(batches):
        before = a.data.clone()
        lp = plain.step(b)["loss"]
        lo = over.step(b)["loss"]
        assert abs(float(lp.detach()) - float(lo.detach())) < 1e-4, i
        if (i + 1) % gas == 0:
            # the three encoder layers were stepped from the hooks, the rest by step(); the gradients were zeroed on the way
            assert float(a.grad.abs().max()) == 0.0 and a._clean_token is not None
            assert (i + 1) // gas == 1 or not torch.equal(before, a.data)        # (warmup_linear's first step has lr 0)
        else:
            assert a._clean_token is None and float(a.grad.abs().max()) > 0.0
            assert torch.equal(before, a.data)
    sp, so = _state(plain), _state(over)
    assert torch.equal(sp["steps"], so["steps"]) and int(so["steps"].max()) == len(batches) // gas
    for k in ("data", "m", "v"):
        assert float((sp[k] - so[k]).abs().max()) < 2e-5, k
    assert float((sp["shadow_t"].float() - so["shadow_t"].float()).abs().max()) < 1e-3


@pytest.mark.gpu
def test_hooks_step_the_layers_and_only_an_armed_backward(dev):
    over, batches = _wrapper(dev, True)
    opt = over.optimizer
    seen = []
    orig = opt._step_range

    def spy(f, c_lo, c_hi, t_lo, t_hi, touched, zero, *stream):
        seen.append((c_lo, c_hi, t_lo, t_hi, touched is None, zero))
        return orig(f, c_lo, c_hi, t_lo, t_hi, touched, zero, *stream)

    opt._step_range = spy
    over.step(batches[0])
    f = opt.fused()
    tables = opt._bucket_tables(f)
    layer_calls = [s for s in seen if s[4]]
    # one call per encoder layer from the hooks (no flags: every backward writes every tensor of a layer), top layer first
    assert [s[:4] for s in layer_calls] == [tables["layer%d" % i][:4] for i in (2, 1, 0)]
    rest = [s for s in seen if not s[4]]
    covered = sorted([s[:2] for s in seen])
    assert covered[0][0] == 0 and covered[-1][1] == f["nc"] and all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))
    assert len(rest) == 2 and all(s[5] for s in seen)          # embeddings below the layers, heads above them
    # a backward nobody announced (a user's own loop: loss.backward(); optimizer.step()) takes the one-pass step
    del seen[:]
    before = over.model.bert.arena.data.clone()
    over.optimizer.zero_grad()
    out = over.model(**batches[1])
    out["loss"].mean().backward()
    assert not seen and torch.equal(before, over.model.bert.arena.data)
    over.optimizer.step()                                       # (the second step of the warm-up: lr > 0)
    assert not seen and not torch.equal(before, over.model.bert.arena.data)
    assert float(over.model.bert.arena.grad.abs().max()) > 0.0   # not zeroed: the caller may still read it
    over.optimizer.zero_grad()
    assert float(over.model.bert.arena.grad.abs().max()) == 0.0


@pytest.mark.gpu
def test_model_wrapper_leaves_the_gradients_readable(dev):
    over, batches = _wrapper(dev, True, zero=False)
    over.step(batches[0])
    a = over.model.bert.arena
    assert over.optimizer._ov["done"] == [] and not over.optimizer._ov["armed"]
    assert a._clean_token is None and float(a.grad.abs().max()) > 0.0
    for p in over.model.bert.bert.encoder.layer[1].parameters():
        assert float(p.grad.abs().max()) > 0.0
    over.step(batches[1])                                       # zero_grad() at the top of the call does the zeroing
    assert float(a.grad.abs().max()) > 0.0


@pytest.mark.gpu
def test_a_backward_after_the_step_is_not_mistaken_for_a_clean_arena(dev):
    over, batches = _wrapper(dev, True)
    over.step(batches[0])
    a = over.model.bert.arena
    assert a._clean_token is not None
    out = over.model(**batches[1])
    out["loss"].mean().backward()                               # writes gradients into the arena the optimizer left zeroed
    assert float(a.grad.abs().max()) > 0.0
    over.optimizer.zero_grad()                                  # must really zero
    assert float(a.grad.abs().max()) == 0.0
