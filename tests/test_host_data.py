"""Host-side data path: vectorised masking against the oracle's restatement of the reference loop."""
import pytest
import torch

from oracle import visualbert_oracle as vo
from visualbert_amd.data import mask_tokens


def test_mask_tokens_matches_reference_loop_draw_for_draw():
    g = torch.Generator().manual_seed(7)
    B, T, V, MASK = 6, 40, 30522, 103
    ids = torch.randint(1000, V, (B, T), generator=g, dtype=torch.int64)
    maskable = torch.ones(B, T, dtype=torch.bool)
    maskable[:, 0] = False                      # [CLS]
    maskable[:, -3:] = False                    # [SEP] + padding
    u = torch.rand(B, T, generator=g, dtype=torch.float64)
    u[0, 5] = 0.15 * 0.8                        # boundaries of the three outcomes, exercised exactly
    u[0, 6] = 0.15 * 0.9
    u[0, 7] = 0.15
    r = torch.randint(0, V, (B, T), generator=g, dtype=torch.int64)
    out, labels = mask_tokens(ids, maskable, V, MASK, uniforms=u, random_ids=r)
    for b in range(B):
        idx = torch.nonzero(maskable[b]).reshape(-1).tolist()       # the reference only sees the real tokens
        toks, labs = vo.random_word_loop([int(ids[b, i]) for i in idx], [float(u[b, i]) for i in idx],
                                         [int(r[b, i]) for i in idx], MASK)
        assert [int(out[b, i]) for i in idx] == toks
        assert [int(labels[b, i]) for i in idx] == labs
        rest = [i for i in range(T) if i not in idx]
        assert all(int(out[b, i]) == int(ids[b, i]) and int(labels[b, i]) == -1 for i in rest)


def test_mask_tokens_rates():
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(1000, 30522, (64, 128), generator=g, dtype=torch.int64)
    out, labels = mask_tokens(ids, torch.ones_like(ids, dtype=torch.bool), 30522, 103, generator=g)
    sel = labels != -1
    rate = sel.float().mean().item()
    assert abs(rate - 0.15) < 0.015
    masked = (out == 103) & sel
    kept = (out == ids) & sel
    assert abs(masked.sum().item() / sel.sum().item() - 0.8) < 0.03
    assert abs(kept.sum().item() / sel.sum().item() - 0.1) < 0.03
    assert torch.equal(labels[sel], ids[sel]) and torch.equal(out[~sel], ids[~sel])


def test_collate_pretraining_matches_reference_loops():
    """caption pairs + ragged region features -> padded batch: the whole-batch collate against the oracle's per-example /
    per-token restatement of bert_data_utils.py:168-261, bert_field.py:79-100, coco_dataset.py:176-181."""
    from visualbert_amd.data import collate_pretraining
    g = torch.Generator().manual_seed(3)
    V, MASK, CLS, SEP, Dv = 30522, 103, 101, 102, 16
    la = [5, 1, 9, 3, 7]
    lb = [4, 0, 2, 6, 0]                               # two single-sentence examples (text_b None)
    ra = [3, 7, 1, 5, 7]
    ids_a = [torch.randint(1000, V, (n,), generator=g) for n in la]
    ids_b = [torch.randint(1000, V, (n,), generator=g) if n else None for n in lb]
    feats = [torch.rand(r, Dv, generator=g) for r in ra]
    correct = [True, False, True, True, False]
    T = max(a + 2 + (b + 1 if b else 0) for a, b in zip(la, lb))
    u = torch.rand(5, T, generator=g, dtype=torch.float64) * 0.4          # ~37 % selected: every branch is taken
    r = torch.randint(0, V, (5, T), generator=g)
    got = collate_pretraining(ids_a, ids_b, correct, feats, V, MASK, CLS, SEP, uniforms=u, random_ids=r, pin=False)
    ex = [dict(ids_a=a.tolist(), ids_b=(b.tolist() if b is not None else None), is_correct=c, features=f)
          for a, b, c, f in zip(ids_a, ids_b, correct, feats)]
    ref = vo.collate_pretraining_loop(ex, u.tolist(), r.tolist(), MASK, CLS, SEP)
    assert set(got) == set(ref)
    for k in ref:
        assert got[k].dtype == ref[k].dtype and torch.equal(got[k], ref[k]), k
    assert int((got["masked_lm_labels"] != -1).sum()) > 3


def test_lr_schedules_match_reference_classes():
    """every _LRSchedule of the reference (optimization.py:37-173), step by step, against values its own classes
    produced (tests/golden/schedules.json, written by oracle/make_golden.py)."""
    import json
    import os
    from visualbert_amd import optimization as opt
    here = os.path.dirname(os.path.abspath(__file__))
    fx = json.load(open(os.path.join(here, "golden", "schedules.json")))
    assert len(fx["cases"]) == 6
    for case in fx["cases"]:
        sch = getattr(opt, case["schedule"])(t_total=fx["t_total"], **case["kwargs"])
        got = [sch.get_lr(s) for s in range(len(case["lr"]))]
        assert max(abs(a - b) for a, b in zip(got, case["lr"])) < 1e-12, case["schedule"]
    assert opt.WarmupLinearSchedule(warmup=0.1, t_total=-1).get_lr(7) == 1.0


def test_from_pretrained_local_archive_and_legacy_names(tmp_path):
    """PreTrainedBertModel.from_pretrained (modeling.py:486-596) on a local directory: bert_config.json + pytorch_model.bin
    with the TF-era LayerNorm.gamma / beta names (:556-568); tensors the archive lacks (the visual tables of a plain BERT
    checkpoint) keep their initialisation; random_initialize=True skips the weights."""
    import json
    from visualbert_amd.modeling import BertConfig, TrainVisualBERTObjective
    cfg = BertConfig(1000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512)
    src = TrainVisualBERTObjective(cfg, "pretraining", visual_embedding_dim=256)
    sd = {}
    for k, v in src.state_dict().items():
        if "_visual" in k or "projection" in k:
            continue                                          # a text-only BERT archive has none of these
        sd[k.replace("LayerNorm.weight", "LayerNorm.gamma").replace("LayerNorm.bias", "LayerNorm.beta")] = v.clone()
    with open(tmp_path / "bert_config.json", "w") as f:
        f.write(cfg.to_json_string())
    torch.save(sd, tmp_path / "pytorch_model.bin")
    assert json.loads(cfg.to_json_string())["hidden_size"] == 128
    torch.manual_seed(5)
    got = TrainVisualBERTObjective.from_pretrained(str(tmp_path), training_head_type="pretraining", visual_embedding_dim=256)
    own, ref = got.state_dict(), src.state_dict()
    for k in ref:
        if k.endswith("projection.bias"):
            continue                                          # zero in both (init_bert_weights)
        if "_visual" in k or "projection" in k:
            assert not torch.equal(own[k], ref[k]), k         # not in the archive: fresh initialisation
        else:
            assert torch.equal(own[k], ref[k]), k
    assert got.cls.predictions.decoder.weight is got.bert.embeddings.word_embeddings.weight     # still tied, still in the arena
    assert got.bert.embeddings.word_embeddings.weight._vb_arena is got.arena
    rnd = TrainVisualBERTObjective.from_pretrained(str(tmp_path), random_initialize=True, training_head_type="pretraining",
                                                   visual_embedding_dim=256)
    assert not torch.equal(rnd.state_dict()["bert.pooler.dense.weight"], ref["bert.pooler.dense.weight"])
    with pytest.raises(FileNotFoundError):
        TrainVisualBERTObjective.from_pretrained("bert-base-uncased", training_head_type="pretraining")
