"""SURVEY.md section 8 row "cfg" (added by the round-3 judge): the reference's configs/*.json and
ModelWrapper.read_and_insert_args (visualbert/models/model_wrapper.py:235-244, train.py:87,200) drop in unchanged.

CPU-only: models are constructed on the meta-free CPU path (no kernel runs); BERT-base construction is ~0.5 GB and a few seconds.
The reference's own config files are read where they lie when /root/reference exists (this container); a config written for
this repository in the same commented-JSON format is always tested."""
import argparse
import glob
import os

import pytest
import torch

from visualbert_amd.model import ModelWrapper, VisualBERTFixedImageEmbedding, load_commented_json
from visualbert_amd.modeling import (PRETRAINED_MODEL_ARCHITECTURES, BertConfig, TrainVisualBERTObjective, resolve_pretrained)
from visualbert_amd.optimization import BertAdam

HERE = os.path.dirname(os.path.abspath(__file__))
SAMPLE = os.path.join(HERE, "configs", "sample-coco-pre-train.json")
REF_CONFIGS = "/root/reference/visualbert/configs"


def _cli(**kw):
    ns = argparse.Namespace(folder="/tmp/run", no_tqdm=False, config=None)
    for k, v in kw.items():
        setattr(ns, k, v)
    return ns


def test_commented_json_reader():
    d = load_commented_json(SAMPLE)
    assert d["bert_model_name"] == "bert-base-uncased" and d["learning_rate"] == 5e-5
    assert d["url_like_string"] == "http://example.invalid/a//b#c"          # comment markers inside a string survive
    assert d["restore_bin"] is None and d["model"]["visual_embedding_dim"] == 2048


def test_read_and_insert_args_semantics():
    """model_wrapper.py:235-244: config first, command line on top, attribute access, model.bert_model_name injected."""
    args = ModelWrapper.read_and_insert_args(_cli(config=SAMPLE, train_batch_size=7), SAMPLE)
    assert args.train_batch_size == 7                      # the command line wins
    assert args.folder == "/tmp/run" and args.config == SAMPLE
    assert args.model.bert_model_name == "bert-base-uncased"
    assert args.model.type == "VisualBERTFixedImageEmbedding"
    assert args.get("fp16", False) is False


def test_known_names_resolve_to_their_architecture(tmp_path):
    cfg, d = resolve_pretrained("bert-base-uncased")
    assert d is None and (cfg.vocab_size, cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads,
                          cfg.intermediate_size) == (30522, 768, 12, 12, 3072)
    cfg, _ = resolve_pretrained("bert-large-uncased")
    assert (cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.intermediate_size) == (1024, 24, 16, 4096)
    assert set(PRETRAINED_MODEL_ARCHITECTURES) == {
        "bert-base-uncased", "bert-large-uncased", "bert-base-cased", "bert-large-cased", "bert-base-multilingual-uncased",
        "bert-base-multilingual-cased", "bert-base-chinese"}                 # modeling.py:44-52
    with pytest.raises(FileNotFoundError):
        resolve_pretrained("no-such-model")
    # a local directory wins over the built-in table, also through cache_dir/<name>
    small = BertConfig(100, hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=128)
    d = tmp_path / "bert-base-uncased"
    d.mkdir()
    (d / "bert_config.json").write_text(small.to_json_string())
    cfg, where = resolve_pretrained("bert-base-uncased", cache_dir=str(tmp_path))
    assert cfg.hidden_size == 128 and cfg.vocab_size == 100 and where == str(d)


def test_from_pretrained_loads_local_weights(tmp_path):
    small = BertConfig(100, hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=128)
    src = TrainVisualBERTObjective(small, "pretraining", visual_embedding_dim=32)
    d = tmp_path / "m"
    d.mkdir()
    (d / "bert_config.json").write_text(small.to_json_string())
    sd = {k.replace("LayerNorm.weight", "LayerNorm.gamma").replace("LayerNorm.bias", "LayerNorm.beta"): v.clone()
          for k, v in src.state_dict().items()}                               # legacy names, modeling.py:556-568
    torch.save(sd, str(d / "pytorch_model.bin"))
    dst = TrainVisualBERTObjective.from_pretrained(str(d), None, None, False, "pretraining", visual_embedding_dim=32)
    for k, v in src.state_dict().items():
        assert torch.equal(dst.state_dict()[k], v), k


def _check_wrapper(args, n_train=10000):
    mw = ModelWrapper(args, n_train)
    assert isinstance(mw.model, VisualBERTFixedImageEmbedding) and isinstance(mw.optimizer, BertAdam)
    cfg = mw.model.bert.config
    assert (cfg.vocab_size, cfg.hidden_size, cfg.num_hidden_layers) == (30522, 768, 12)
    assert mw.model.bert.bert.embeddings.projection.weight.shape == (768, args.model.visual_embedding_dim)
    steps = int(n_train / args.train_batch_size / args.gradient_accumulation_steps) * args.num_train_epochs
    assert mw.num_train_optimization_steps == steps                             # model_wrapper.py:113-115
    assert all("pooler" not in n for n in mw.optimizer_param_names)             # model_wrapper.py:106
    if args.model.get("special_visual_initialize"):
        e = mw.model.bert.bert.embeddings
        assert torch.equal(e.position_embeddings_visual.weight, e.position_embeddings.weight)
    return mw


def test_sample_config_constructs_model_and_optimizer():
    args = ModelWrapper.read_and_insert_args(_cli(config=SAMPLE), SAMPLE)
    mw = _check_wrapper(args)
    assert mw.model.training_head_type == "pretraining"
    assert sum(p.numel() for p in mw.model.parameters()) == 112074812


def _reference_configs():
    return sorted(glob.glob(os.path.join(REF_CONFIGS, "*", "*.json")))


@pytest.mark.skipif(not os.path.isdir(REF_CONFIGS), reason="the reference tree is not on this machine")
@pytest.mark.parametrize("path", _reference_configs(), ids=lambda p: "/".join(p.split("/")[-2:]))
def test_reference_configs_drop_in(path):
    """every config the reference ships: read by read_and_insert_args, and -- for the model type this package builds -- model +
    optimizer constructed from it, unchanged.  The VCR configs name VisualBERTDetector (detectron backbone inside the model:
    out of scope) and must fail loudly, not build something else."""
    args = ModelWrapper.read_and_insert_args(_cli(config=path), path)
    assert args.model.bert_model_name == args.bert_model_name == "bert-base-uncased"
    if args.model.type != "VisualBERTFixedImageEmbedding":
        with pytest.raises(NotImplementedError):
            ModelWrapper(args, 1000)
        return
    mw = _check_wrapper(args)
    assert mw.model.training_head_type == args.model.training_head_type


def test_restore_checkpoint_pretrained_reports_and_refuses_a_checkpoint_that_matches_nothing(tmp_path, caplog):
    """model_wrapper.py:201-221 prints Skipped / Successfully loaded / Part load failed per key; here the same pass strips a
    DataParallel "module." prefix, logs the counts and RAISES when no tensor matched (a mistyped checkpoint must not silently
    train from random weights -- ADVICE r04)."""
    import logging
    small = BertConfig(100, hidden_size=128, num_hidden_layers=1, num_attention_heads=2, intermediate_size=128)
    src = VisualBERTFixedImageEmbedding(config=small, training_head_type="nlvr", visual_embedding_dim=32)
    dst = VisualBERTFixedImageEmbedding(config=small, training_head_type="nlvr", visual_embedding_dim=32)
    mw = ModelWrapper.__new__(ModelWrapper)                       # the loader needs the model only
    mw.model = dst
    mw._after_weights_changed = lambda: None
    good = {"module." + k: v.clone() for k, v in src.state_dict().items()}
    some = next(k for k in good if k.endswith("query.weight"))
    good["module.not.a.parameter"] = torch.zeros(3)
    good[some] = torch.zeros(5, 5)                                # one shape mismatch
    torch.save(good, str(tmp_path / "good.th"))
    with caplog.at_level(logging.INFO, logger="visualbert_amd.model"):
        loaded, unknown, mismatch = mw.restore_checkpoint_pretrained(str(tmp_path / "good.th"))
    assert unknown == ["not.a.parameter"] and mismatch == [some[len("module."):]]
    assert len(loaded) == len(src.state_dict()) - 1
    assert any("Skipped: not.a.parameter" in r.message for r in caplog.records)
    assert any("Part load failed" in r.message for r in caplog.records)
    for k, v in src.state_dict().items():
        if k != some[len("module."):]:
            assert torch.equal(dst.state_dict()[k], v), k
    torch.save({"totally.different": torch.zeros(2)}, str(tmp_path / "bad.th"))
    with pytest.raises(RuntimeError, match="no tensor of the checkpoint matches"):
        mw.restore_checkpoint_pretrained(str(tmp_path / "bad.th"))
