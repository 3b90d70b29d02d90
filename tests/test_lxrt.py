"""The sibling model's cross-modality blocks (visualbert_amd/lxrt.py) against the REAL reference's outputs and gradients
(tests/golden/micro_lxrt.npz from unsupervised_visualbert/src/lxrt/modeling.py: LXRTXLayer :660-712 fed by a
VisualFeatEncoder :715-747).  fp32 kernels at fp32 tolerances; bf16 kernels measured (recorded) and bounded at 1.5 x."""
import numpy as np
import pytest
import torch

from golden_util import maxdiff, record
from test_oracle_golden import lxrt_case

pytestmark = pytest.mark.gpu

# bf16: max |d(out)|, worst / median per-tensor gradient relative L2 -- 1.5 x the values measured on MI355X
# (profiles/r02_parity_small.json: outputs 2.9e-2 at |out| <= 3.9, i.e. after three LayerNorms; gradients median 5.3e-3, worst 1.39e-2)
BF16_BOUNDS = dict(out=4.4e-2, grad_median=8e-3, grad_worst=2.1e-2)


def build(dev, dtype):
    from visualbert_amd import lxrt
    from visualbert_amd.modeling import BertConfig
    cfg, sd, x, wl, wv, g = lxrt_case()
    bc = BertConfig(cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                    num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size)
    layer = lxrt.LXRTXLayer(bc)
    enc = lxrt.VisualFeatEncoder(bc, visual_feat_dim=x["feats"].size(-1), visual_pos_dim=4)
    missing = layer.load_state_dict({k: v for k, v in sd.items() if not k.startswith("enc.")}, strict=True)
    enc.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("enc.")}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys          # the reference's state-dict keys, exactly
    layer, enc = layer.to(dev).eval(), enc.to(dev).eval()
    xin = {k: v.to(dev) for k, v in x.items()}
    lang = xin["lang"].to(dtype).requires_grad_(True)
    feats = xin["feats"].to(dtype).requires_grad_(True)
    visn_in = enc((feats, xin["boxes"].to(dtype)))
    lo, vo_ = layer(lang, xin["lang_ext_mask"], visn_in, xin["visn_ext_mask"])
    loss = (lo.float() * wl.to(dev)).sum() + (vo_.float() * wv.to(dev)).sum()
    loss.backward()
    grads = {n: p.grad.detach().float().cpu() for n, p in layer.named_parameters()}
    grads.update({"enc." + n: p.grad.detach().float().cpu() for n, p in enc.named_parameters()})
    return g, visn_in, lo, vo_, lang, feats, grads


def test_fp32_lxrt_blocks_match_reference_golden(dev):
    g, visn_in, lo, vo_, lang, feats, grads = build(dev, torch.float32)
    assert maxdiff(visn_in.detach().cpu(), g["visn_encoded"]) < 1e-4
    assert maxdiff(lo.detach().cpu(), g["lang_out"]) < 1e-4 and maxdiff(vo_.detach().cpu(), g["visn_out"]) < 1e-4
    assert maxdiff(lang.grad.cpu(), g["grad_in/lang"]) < 1e-3 and maxdiff(feats.grad.cpu(), g["grad_in/feats"]) < 1e-3
    for n, gr in grads.items():
        ref = torch.as_tensor(g["grad/" + n])
        if n.endswith("key.bias"):              # d/d(key bias) is identically 0 (softmax shift invariance): rounding noise on both sides
            assert float(gr.norm()) < 1e-5, n
            continue
        assert float((gr - ref).norm()) <= 2e-3 * float(ref.norm()) + 1e-6, n


def test_bf16_lxrt_blocks_measured_against_fp32_reference(dev):
    g, visn_in, lo, vo_, lang, feats, grads = build(dev, torch.bfloat16)
    out_err = max(maxdiff(lo.detach().float().cpu(), g["lang_out"]), maxdiff(vo_.detach().float().cpu(), g["visn_out"]))
    rels = sorted(float((gr - torch.as_tensor(g["grad/" + n])).norm()) / (float(torch.as_tensor(g["grad/" + n]).norm()) + 1e-12)
                  for n, gr in grads.items() if not n.endswith("key.bias"))
    rec = dict(max_dout=out_err, out_absmax=float(np.abs(g["lang_out"]).max()), grad_rel_l2_median=rels[len(rels) // 2],
               grad_rel_l2_worst=rels[-1])
    record("bf16_lxrt", "micro_lxrt", rec)
    print("bf16 lxrt: %s" % rec)
    assert out_err <= BF16_BOUNDS["out"], rec
    assert rec["grad_rel_l2_median"] <= BF16_BOUNDS["grad_median"] and rec["grad_rel_l2_worst"] <= BF16_BOUNDS["grad_worst"], rec


@pytest.mark.parametrize("mode", ["fp32", "bf16x3", "bf16"])
@pytest.mark.parametrize("tag", ["style", "lrx"])
def test_lxrt_encoder_matches_reference_golden(dev, tag, mode):
    """LXRTEncoder (lxrt/modeling.py:769-905) at BERT-base width (2 language, 1 relational, 2 cross-modality layers; 20 tokens
    + 36 regions of 2048-d features) against the REAL reference's forward and gradients: the fp32 kernels and the
    split-operand bf16x3 mode at fp32 bounds, the bf16 kernels measured and bounded."""
    from golden_util import sub
    from test_oracle_golden import lxrt_encoder_case
    from visualbert_amd import lxrt, ops
    from visualbert_amd.modeling import BertConfig
    if dev.type != "cuda":
        pytest.skip("BERT-base width: GPU only")
    cfg, sd, x, wl, wv, g, (n_l, n_r, n_x) = lxrt_encoder_case(tag)
    bc = BertConfig(cfg.vocab_size, hidden_size=cfg.hidden_size, num_hidden_layers=cfg.num_hidden_layers,
                    num_attention_heads=cfg.num_attention_heads, intermediate_size=cfg.intermediate_size)
    enc = lxrt.LXRTEncoder(bc, l_layers=n_l, x_layers=n_x, r_layers=n_r, visualbert_style=(tag == "style"),
                           visual_feat_dim=x["feats"].size(-1))
    res = enc.load_state_dict(sd, strict=True)                            # the reference's state-dict keys, exactly
    assert not res.missing_keys and not res.unexpected_keys
    enc = enc.to(dev).eval()
    dt = torch.bfloat16 if mode == "bf16" else torch.float32
    xin = {k: v.to(dev) for k, v in x.items()}
    lang = xin["lang"].to(dt).requires_grad_(True)
    feats = xin["feats"].to(dt).requires_grad_(True)
    with ops.x3_scope(mode == "bf16x3"):
        lo, vo_ = enc(lang, xin["lang_ext_mask"], (feats, xin["boxes"].to(dt)), xin["visn_ext_mask"])
        loss = (lo.float() * wl.to(dev)).sum() + (vo_.float() * wv.to(dev)).sum()
    loss.backward()
    out_err = max(maxdiff(sub(lo.detach().float().cpu()), g[tag + "/lang_out_sub"]),
                  maxdiff(sub(vo_.detach().float().cpu()), g[tag + "/visn_out_sub"]))
    rels = {}
    for n, p in enc.named_parameters():
        ref = torch.as_tensor(g[tag + "/grad_sub/" + n])
        if n.endswith("key.bias") or float(ref.norm()) < 1e-7:
            continue
        rels[n] = float((sub(p.grad.detach().float().cpu()) - ref).norm()) / float(ref.norm())
    worst = max(rels, key=rels.get)
    record("lxrt_encoder_" + mode, tag, dict(max_dout=out_err, out_absmax=float(g[tag + "/out_absmax"]),
                                             grad_rel_l2_worst=rels[worst], grad_rel_l2_worst_name=worst))
    if mode == "bf16":
        # bf16 kernels, measured on MI355X (profiles/r03_parity_small.json) and bounded at 1.5 x: style -- max |dout| 2.9e-2 at
        # |out| <= 4.7, worst gradient 2.0e-2; lrx -- 3.3e-2 at |out| <= 3.7, worst gradient 0.116 (a key projection of the
        # last cross-modality layer: its gradient is a difference of nearly equal terms, softmax shift invariance)
        bound_out, bound_grad = (4.3e-2, 3.0e-2) if tag == "style" else (5.0e-2, 0.175)
        assert out_err <= bound_out and rels[worst] <= bound_grad, (out_err, worst, rels[worst])
    else:
        assert out_err < 1e-4, out_err
        assert maxdiff(sub(lang.grad.float().cpu()), g[tag + "/grad_in_lang_sub"]) < 1e-3
        assert rels[worst] <= 2e-3, (worst, rels[worst])
