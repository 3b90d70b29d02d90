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
    with pytest.raises(FileNotFoundError):                    # neither a directory nor a name of the reference's archive map
        TrainVisualBERTObjective.from_pretrained("bert-base-unknown", training_head_type="pretraining")
    # (a known name resolves to its architecture offline: tests/test_config_dropin.py)


def test_region_feature_store_reads_the_reference_file_layout(tmp_path):
    """one .npy of float32 [regions, Dv] per image, named as the reference names them (dataloaders/coco_dataset.py:144-155);
    a batch lands zero-padded in ONE pre-padded slab with image_dim_variable = region counts, and a slot is reusable."""
    import numpy as np
    from visualbert_amd.data import RegionFeatureStore
    rng = np.random.RandomState(0)
    ids, arrays = [9, 57870, 139], {}
    for i, r in zip(ids, (5, 3, 8)):
        arrays[i] = rng.rand(r, 16).astype(np.float32)
        np.save(str(tmp_path / ("COCO_train2014_%012d.npy" % i)), arrays[i])
    store = RegionFeatureStore(str(tmp_path), "train")
    out = store.read_batch(ids, pin=False)
    assert tuple(out["image_feat_variable"].shape) == (3, 8, 16) and out["image_dim_variable"].tolist() == [5, 3, 8]
    for b, i in enumerate(ids):
        r = arrays[i].shape[0]
        assert np.array_equal(out["image_feat_variable"][b, :r].numpy(), arrays[i])
        assert float(out["image_feat_variable"][b, r:].abs().sum()) == 0.0
    class Ev(object):                                   # the event FeatureStager.stage() returned for this slab's last DMA
        waited = 0

        def synchronize(self):
            # the refill must not have started yet: the slab still holds the previous batch
            assert np.array_equal(out["image_feat_variable"][0, :arrays[ids[0]].shape[0]].numpy(), arrays[ids[0]])
            Ev.waited += 1
    again = store.read_batch(list(reversed(ids)), regions=8, out=out, pin=False, wait=Ev())   # same slot, other images
    assert Ev.waited == 1
    assert again["image_feat_variable"].data_ptr() == out["image_feat_variable"].data_ptr()
    assert again["image_dim_variable"].tolist() == [8, 3, 5]
    assert float(again["image_feat_variable"][1, 3:].abs().sum()) == 0.0                # stale rows of the longer image are cleared
    fixed = store.read_batch(ids, regions=4, pin=False)                                  # truncation to a fixed region count
    assert fixed["image_dim_variable"].tolist() == [4, 3, 4]


def test_init_bert_weights_and_special_initialize():
    """a16: init_bert_weights (modeling.py:473-484: N(0, initializer_range) for Linear / Embedding weights, zero biases,
    LayerNorm (1, 0)) and special_intialize (modeling.py:1191-1196: the *_visual tables are COPIES of the text tables)."""
    from visualbert_amd.modeling import BertConfig, TrainVisualBERTObjective
    torch.manual_seed(0)
    cfg = BertConfig(3000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512)
    m = TrainVisualBERTObjective(cfg, "pretraining", visual_embedding_dim=256)
    sd = m.state_dict()
    for n, t in sd.items():
        if n.endswith("LayerNorm.weight"):
            assert bool((t == 1).all()), n
        elif n.endswith("LayerNorm.bias") or n.endswith(".bias"):
            assert bool((t == 0).all()), n
        elif t.numel() >= 16384:
            assert abs(float(t.std()) - cfg.initializer_range) < 0.1 * cfg.initializer_range, (n, float(t.std()))
            assert abs(float(t.mean())) < 5e-4, n
    e = m.bert.embeddings
    assert not torch.equal(e.token_type_embeddings_visual.weight, e.token_type_embeddings.weight)    # independent draws ...
    e.special_intialize()
    assert torch.equal(e.token_type_embeddings_visual.weight, e.token_type_embeddings.weight)        # ... until copied
    assert torch.equal(e.position_embeddings_visual.weight, e.position_embeddings.weight)
    assert e.token_type_embeddings_visual.weight.data_ptr() != e.token_type_embeddings.weight.data_ptr()
    with torch.no_grad():
        e.token_type_embeddings.weight.add_(1.0)            # a copy, not an alias
    assert not torch.equal(e.token_type_embeddings_visual.weight, e.token_type_embeddings.weight)
    assert m.cls.predictions.decoder.weight is m.bert.embeddings.word_embeddings.weight               # tied (modeling.py:414)


# ---- pinned to the REAL reference's host-side text code (tests/golden/host_text.json, oracle/make_golden_host.py) ------------
def _host_fixture():
    import json
    import os
    from golden_util import GOLDEN_DIR
    with open(os.path.join(GOLDEN_DIR, "host_text.json"), encoding="utf-8") as f:
        return json.load(f)


def test_wordpiece_encoder_matches_reference_tokenizer():
    """BertTokenizer.tokenize / convert_tokens_to_ids of the reference (tokenization.py:74-166) on 18 sentences that hit every
    branch: accents, CJK, control / format characters, Unicode spaces, the 100-character limit, unmatched remainders."""
    from visualbert_amd.tokenization import WordPieceEncoder
    fx = _host_fixture()
    vocab = {t: i for i, t in enumerate(fx["vocab"])}
    for key, lower in (("tokenize", True), ("tokenize_cased", False)):
        enc = WordPieceEncoder(vocab, do_lower_case=lower)
        for rep in range(2):                                  # second round: served from the per-word memo
            for case in fx[key]:
                toks = enc.tokenize(case["text"])
                assert toks == case["tokens"], (case["text"], toks, case["tokens"])
                assert enc.convert_tokens_to_ids(toks) == case["ids"]
                assert enc.encode(case["text"]) == case["ids"]
    assert enc.convert_ids_to_tokens([2, 3, 4]) == ["[CLS]", "[SEP]", "[MASK]"]


def _draw_arrays(cases, T):
    """the reference's recorded draws, placed at each token's FINAL position ([CLS] a [SEP] b [SEP])"""
    B = len(cases)
    u = torch.ones((B, T), dtype=torch.float64)               # 1.0 = never selected (special tokens, padding)
    r = torch.zeros((B, T), dtype=torch.int64)
    for b, c in enumerate(cases):
        la = len(c["ids_a"])
        pos = list(range(1, 1 + la)) + (list(range(la + 2, la + 2 + len(c["ids_b"]))) if c["ids_b"] else [])
        assert len(pos) == len(c["draws"])
        for p, d in zip(pos, c["draws"]):
            u[b, p] = d["u"]
            r[b, p] = max(d["choice_id"], 0)
    return u, r


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_masking_and_features_reproduce_the_reference_on_its_own_draws(seed):
    """random_word (fine_tuning.py:272-308) + convert_one_example_to_features_pretraining (bert_data_utils.py:168-247) were
    RUN from the reference with random.random / random.choice recorded; on those draws both the oracle's restatement
    (collate_pretraining_loop) and the vectorised product path (collate_pretraining -> mask_tokens) must give the reference's
    input ids, segment ids, mask and LM labels element for element."""
    from visualbert_amd.data import collate_pretraining
    fx = _host_fixture()
    vocab = {t: i for i, t in enumerate(fx["vocab"])}
    cases = [c for c in fx["pretraining_features"] if c["seed"] == seed]
    assert len(cases) == 5
    prob = cases[0]["probability"]
    T = max(len(c["input_ids"]) for c in cases)
    u, r = _draw_arrays(cases, T)
    feats = [torch.rand(3, 8) for _ in cases]
    examples = [dict(ids_a=c["ids_a"], ids_b=c["ids_b"], is_correct=c["is_correct"], features=f) for c, f in zip(cases, feats)]
    ora = vo.collate_pretraining_loop(examples, u.tolist(), r.tolist(), vocab["[MASK]"], vocab["[CLS]"], vocab["[SEP]"], prob)
    prod = collate_pretraining([torch.tensor(c["ids_a"]) for c in cases],
                               [torch.tensor(c["ids_b"]) if c["ids_b"] else None for c in cases],
                               [c["is_correct"] for c in cases], feats, len(vocab), vocab["[MASK]"], vocab["[CLS]"],
                               vocab["[SEP]"], probability=prob, uniforms=u, random_ids=r, pin=False)
    for name, got in (("oracle", ora), ("product", prod)):
        for b, c in enumerate(cases):
            n = len(c["input_ids"])
            assert got["bert_input_ids"][b, :n].tolist() == c["input_ids"], (name, b)
            assert got["bert_input_mask"][b, :n].tolist() == c["input_mask"], (name, b)
            assert got["bert_input_type_ids"][b, :n].tolist() == c["input_type_ids"], (name, b)
            assert got["masked_lm_labels"][b, :n].tolist() == c["lm_label_ids"], (name, b)
            assert got["bert_input_ids"][b, n:].tolist() == [0] * (T - n)                      # padded with 0 ...
            assert got["masked_lm_labels"][b, n:].tolist() == [-1] * (T - n)                  # ... and -1 (bert_data_utils.py:249-256)
            assert int(got["is_random_next"][b]) == int(c["is_correct"])
    # the fixture really exercises all three outcomes of the masking rule
    flat = [d for c in fx["pretraining_features"] for d in c["draws"]]
    assert any(d["choice_id"] >= 0 for d in flat) and any(d["u"] < 0.15 * 0.8 for d in flat) and any(d["u"] >= 0.5 for d in flat)
