"""Boundary callers of the hot path, restated without AllenNLP (which cannot be installed here):

  VisualBERTFixedImageEmbedding   visualbert/models/model.py:191-301  (task model: builds image_mask from
                                  image_dim_variable, forwards the batch kwargs to TrainVisualBERTObjective)
  ModelWrapper                    visualbert/models/model_wrapper.py:34-147 (step(): zero_grad -> forward ->
                                  loss.mean() -> backward -> optimizer.step; optimizer param groups)

Same keyword names, same `args` keys (train_batch_size, learning_rate, warmup_proportion,
num_train_epochs, gradient_accumulation_steps, fp16, model.{...}).  One process per GPU: where the
reference wraps the model in nn.DataParallel (model_wrapper.py:146), this wrapper takes an optional
visualbert_amd.parallel.DataParallelGradSync that all-reduces the flat gradient arena over RCCL.
"""
import torch
from torch import nn

from . import ops
from .modeling import BertConfig, TrainVisualBERTObjective
from .optimization import BertAdam


class VisualBERTFixedImageEmbedding(nn.Module):
    """models/model.py:191-301 without the AllenNLP Model base / metrics objects."""

    def __init__(self, config=None, bert_model_name=None, training_head_type="pretraining", visual_embedding_dim=2048,
                 hard_cap_seq_len=None, cut_first="text", embedding_strategy="plain", bypass_transformer=False,
                 random_initialize=True, output_attention_weights=False, special_visual_initialize=True,
                 compute_dtype=torch.float32, class_embs=True, cnn_loss_ratio=0.0, vocab=None, text_only=False):
        super(VisualBERTFixedImageEmbedding, self).__init__()
        kw = dict(visual_embedding_dim=visual_embedding_dim, hard_cap_seq_len=hard_cap_seq_len, cut_first=cut_first,
                  embedding_strategy=embedding_strategy, bypass_transformer=bypass_transformer,
                  output_attention_weights=output_attention_weights, compute_dtype=compute_dtype)
        if config is None and bert_model_name is not None:
            # models/model.py:213-223: TrainVisualBERTObjective.from_pretrained(bert_model_name, ...): a name of the reference's
            # archive map ("bert-base-uncased" in every shipped config) or a local directory (modeling.resolve_pretrained)
            self.bert = TrainVisualBERTObjective.from_pretrained(bert_model_name, None, None, random_initialize,
                                                                 training_head_type, **kw)
        else:
            self.bert = TrainVisualBERTObjective(config if config is not None else BertConfig(30522), training_head_type, **kw)
        self.text_only = text_only
        if special_visual_initialize:
            self.bert.bert.embeddings.special_intialize()            # models/model.py:224-225
        self.training_head_type = training_head_type
        self.cnn_loss_ratio = cnn_loss_ratio
        self._arange_cache = {}

    def forward(self, bert_input_ids, bert_input_mask, bert_input_type_ids, image_dim_variable=None,
                image_feat_variable=None, image_text_alignment=None, visual_embeddings_type=None, label=None,
                flickr_position=None, masked_lm_labels=None, is_random_next=None, output_all_encoded_layers=False):
        if image_feat_variable is not None:
            # models/model.py:262-268: image_mask = arange(R) < image_dim_variable  (int64, bit-exact)
            # (one launch: the arange is cached per (R, device) and the comparison writes int64 directly -- at the reference's own batch
            # sizes the step is a chain of small launches, every ATen one counts)
            R = image_feat_variable.size(-2)
            key = (R, image_feat_variable.device)
            ar = self._arange_cache.get(key)
            if ar is None:
                ar = self._arange_cache[key] = torch.arange(R, device=image_feat_variable.device)
            ar = ar.expand(*image_feat_variable.size()[:-1])
            dim = image_dim_variable
            if dim.dim() < ar.dim():
                dim = dim.unsqueeze(-1)
            image_mask = torch.lt(ar, dim, out=torch.empty(ar.shape, dtype=torch.long, device=ar.device))
        else:
            image_mask = None
        output_dict = self.bert(
            input_ids=bert_input_ids, token_type_ids=bert_input_type_ids, input_mask=bert_input_mask,
            visual_embeddings=image_feat_variable, position_embeddings_visual=None, image_mask=image_mask,
            visual_embeddings_type=visual_embeddings_type, image_text_alignment=image_text_alignment, label=label,
            flickr_position=flickr_position, masked_lm_labels=masked_lm_labels, is_random_next=is_random_next,
            output_all_encoded_layers=output_all_encoded_layers)
        output_dict["cnn_regularization_loss"] = None
        return output_dict


def _load_flexible(model, state_dict, strict_first=True, report=None):
    """utils/pytorch_misc.py:246-265: try a full load, then fall back to copying the tensors whose names match.
    -> (loaded, skipped_unknown, shape_mismatch) name lists; `report` (a callable taking one line) gets the reference's per-key
    messages ("Skipped" / "Part load failed", models/model_wrapper.py:201-221 via utils/pytorch_misc.py)."""
    if strict_first:
        try:
            model.load_state_dict(state_dict)
            return list(state_dict.keys()), [], []
        except RuntimeError:
            pass
    own_state = model.state_dict()
    loaded, unknown, mismatch = [], [], []
    with torch.no_grad():
        for name, param in state_dict.items():
            if name not in own_state:
                unknown.append(name)
                if report:
                    report("Skipped: " + name)
                continue
            if isinstance(param, torch.nn.Parameter):
                param = param.data
            if own_state[name].shape == param.shape:
                own_state[name].copy_(param)
                loaded.append(name)
            else:
                mismatch.append(name)
                if report:
                    report("Part load failed: %s (checkpoint %s, model %s)" % (name, tuple(param.shape), tuple(own_state[name].shape)))
    return loaded, unknown, mismatch


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v


def load_commented_json(path):
    """The reference reads its configs with `commentjson` (models/model_wrapper.py:236-239; not installable here): JSON with
    `//` and `#` line comments (also /* */ blocks) and a trailing comma tolerated.  Comment markers inside strings are kept."""
    import json
    with open(path, "r", encoding="utf-8") as f:
        text = f.read()
    out, i, n, in_str = [], 0, len(text), False
    while i < n:
        c = text[i]
        if in_str:
            out.append(c)
            if c == "\\" and i + 1 < n:
                out.append(text[i + 1])
                i += 1
            elif c == '"':
                in_str = False
        elif c == '"':
            in_str = True
            out.append(c)
        elif c == "#" or text.startswith("//", i):
            while i < n and text[i] != "\n":
                i += 1
            continue
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            continue
        else:
            out.append(c)
        i += 1
    text = "".join(out)
    # trailing commas before a closing bracket (outside strings: the comments are gone, so scan once more)
    out, in_str, i, n = [], False, 0, len(text)
    while i < n:
        c = text[i]
        if in_str:
            out.append(c)
            if c == "\\" and i + 1 < n:
                out.append(text[i + 1])
                i += 1
            elif c == '"':
                in_str = False
        elif c == '"':
            in_str = True
            out.append(c)
        elif c == ",":
            j = i + 1
            while j < n and text[j] in " \t\r\n":
                j += 1
            if j < n and text[j] in "}]":
                i += 1
                continue
            out.append(c)
        else:
            out.append(c)
        i += 1
    return json.loads("".join(out))


def _attrify(x):
    if isinstance(x, dict):
        return AttrDict((k, _attrify(v)) for k, v in x.items())
    if isinstance(x, list):
        return [_attrify(v) for v in x]
    return x


class ModelWrapper(object):
    """models/model_wrapper.py:34-147, one process per GPU."""

    #: `model.type` values of the reference's configs that this package builds (VisualBERTDetector carries a detectron
    #: backbone: SURVEY.md section 2, out of scope)
    MODEL_TYPES = ("VisualBERTFixedImageEmbedding",)

    def __init__(self, args, train_dataset_length, model=None, grad_sync=None, device=None):
        self.scheduler = None
        self.args = args if isinstance(args, AttrDict) else _attrify(dict(args))
        self.args.gradient_accumulation_steps = self.args.get("gradient_accumulation_steps", 1)     # model_wrapper.py:38-39
        self.args.fp16 = self.args.get("fp16", False)
        self.device = device
        if model is None:
            model = self.initialize_model(self.args)
        self.model = model.to(device) if device is not None else model
        self.grad_sync = grad_sync
        self.initialize_opimizer(self.args, train_dataset_length)
        if self.args.get("restore_bin", None):
            self.restore_checkpoint_pretrained(self.args.restore_bin)                                 # train.py:203-204
        self.global_step = 0
        self.called_time = 0

    @staticmethod
    def read_and_insert_args(args, confg):
        """models/model_wrapper.py:235-244: the (commented) JSON config, overridden by the command line's attributes, as an
        attribute dictionary; `model.bert_model_name` follows the top-level `bert_model_name`."""
        config_json = load_commented_json(confg)
        dict_args = dict(args) if isinstance(args, dict) else dict(vars(args))
        config_json.update(dict_args)
        args = _attrify(config_json)
        args.model.bert_model_name = args.bert_model_name
        return args

    def initialize_model(self, args):
        """models/model_wrapper.py:141-146 without AllenNLP's registry: `args.model` = {"type": <registered name>, **kwargs}.
        `fp16: true` selects the bf16 kernels (the reference halves the model); the extra key `compute_dtype`
        ("fp32" | "bf16" | "bf16x3"; not in the reference's configs) names a mode directly."""
        m = dict(args.get("model", {}))
        kind = m.pop("type", "VisualBERTFixedImageEmbedding")
        if kind not in self.MODEL_TYPES:
            raise NotImplementedError("visualbert_amd builds model types %s; %r (detector backbone in the model) is out of "
                                      "scope" % (list(self.MODEL_TYPES), kind))
        modes = {"fp32": torch.float32, "bf16": torch.bfloat16, "bf16x3": "bf16x3"}
        dtype = modes[args.get("compute_dtype")] if args.get("compute_dtype") else \
            (torch.bfloat16 if args.get("fp16", False) else torch.float32)
        # the reference's constructor defaults (models/model.py:192-206), which its configs rely on
        m.setdefault("special_visual_initialize", False)
        m.setdefault("training_head_type", "")
        m.setdefault("visual_embedding_dim", 512)
        m.setdefault("random_initialize", False)
        m.setdefault("bert_model_name", "bert-base-uncased")
        return VisualBERTFixedImageEmbedding(compute_dtype=dtype, **m)

    def train(self):
        self.model.train()

    def eval(self):
        self.model.eval()

    def initialize_opimizer(self, args, train_dataset_length):            # (sic) models/model_wrapper.py:100
        param_optimizer = [n for n in self.model.named_parameters() if "pooler" not in n[0]]
        self.optimizer_param_names = [n for n, _ in param_optimizer]
        no_decay = ["bias", "LayerNorm.bias", "LayerNorm.weight"]
        groups = [
            {"params": [p for n, p in param_optimizer if not any(nd in n for nd in no_decay)], "weight_decay": 0.01},
            {"params": [p for n, p in param_optimizer if any(nd in n for nd in no_decay)], "weight_decay": 0.0}]
        steps = int(train_dataset_length / args.train_batch_size / args.get("gradient_accumulation_steps", 1)) * \
            args.get("num_train_epochs", 1)
        self.num_train_optimization_steps = steps
        self.optimizer = BertAdam(groups, lr=args.learning_rate, warmup=args.warmup_proportion, t_total=steps)

    # -- checkpoints: the reference's file names and dictionary keys (models/model_wrapper.py:150-221,
    #    utils/pytorch_misc.py:110-330), so that a run can be resumed by either side ------------------------------------
    def state_dict(self):
        return {"model": self.model.state_dict(), "optimizer": self.optimizer.state_dict()}

    def load_state_dict(self, state_dict_to_load):
        _load_flexible(self.model, state_dict_to_load["model"])
        self.optimizer.load_state_dict(state_dict_to_load["optimizer"])
        self._after_weights_changed()

    def _after_weights_changed(self):
        arena = getattr(getattr(self.model, "bert", None), "arena", None)
        if arena is not None and arena.data.is_cuda:
            arena.refresh_shadows()                         # bf16 / transposed copies follow the fp32 masters

    def save_checkpoint(self, serialization_dir, epoch, val_metric_per_epoch, is_best=False):
        import os
        import shutil
        assert serialization_dir
        model_path = os.path.join(serialization_dir, "model_state_epoch_{}.th".format(epoch))
        torch.save(self.model.state_dict(), model_path)
        torch.save({"epoch": epoch, "val_metric_per_epoch": val_metric_per_epoch,
                    "optimizer": self.optimizer.state_dict()},
                   os.path.join(serialization_dir, "training_state_epoch_{}.th".format(epoch)))
        if is_best:
            shutil.copyfile(model_path, os.path.join(serialization_dir, "best.th"))

    def save_checkpoint_step(self, serialization_dir, step, epoch, is_best=False):
        import os
        assert serialization_dir
        torch.save(self.model.state_dict(),
                   os.path.join(serialization_dir, "model_step_{}_epoch_{}.th".format(step, epoch)))
        torch.save({"step": step, "epoch": epoch, "val_metric_per_epoch": None, "optimizer": self.optimizer.state_dict()},
                   os.path.join(serialization_dir, "training_step_{}_epoch_{}.th".format(step, epoch)))

    def restore_checkpoint(self, serialization_dir, epoch_to_load=None):
        """resume from the newest end-of-epoch checkpoint (or, when there is none, the newest step checkpoint) of a
        serialization directory; returns (epoch to continue with, val_metric_per_epoch) -- (0, []) if there is none."""
        import os
        import re
        files = os.listdir(serialization_dir) if serialization_dir and os.path.isdir(serialization_dir) else []
        epochs = sorted(int(m.group(1)) for m in (re.fullmatch(r"model_state_epoch_([0-9]+)\.th", f) for f in files) if m)
        if epoch_to_load is not None:
            epochs = [e for e in epochs if e == int(epoch_to_load)]
        if epochs:
            e = epochs[-1]
            pair = ("model_state_epoch_%d.th" % e, "training_state_epoch_%d.th" % e)
        else:
            steps = sorted((int(m.group(2)), int(m.group(1)), f) for m, f in
                           ((re.fullmatch(r"model_step_([0-9]+)_epoch_([0-9]+)\.th", f), f) for f in files) if m)
            if not steps:
                return 0, []
            ep, st, f = steps[-1]
            pair = (f, "training_step_%d_epoch_%d.th" % (st, ep))
        model_state = torch.load(os.path.join(serialization_dir, pair[0]), map_location="cpu")
        training_state = torch.load(os.path.join(serialization_dir, pair[1]), map_location="cpu")
        # a checkpoint saved from the reference's nn.DataParallel wrapper carries a "module." prefix on every key
        # (models/model_wrapper.py:146, 163-169); _load_flexible tolerates missing / extra keys like the reference's
        # restore (utils/pytorch_misc.py:246-265)
        model_state = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in model_state.items()}
        _load_flexible(self.model, model_state)
        self.optimizer.load_state_dict(training_state["optimizer"])
        self._after_weights_changed()
        epoch = training_state["epoch"]
        epoch_to_return = (epoch if isinstance(epoch, int) else int(str(epoch).split(".")[0])) + 1
        return epoch_to_return, training_state.get("val_metric_per_epoch", [])

    def restore_checkpoint_pretrained(self, restore_bin):
        """copy every tensor of a saved state dict whose name exists here (models/model_wrapper.py:201-221)."""
        import logging
        log = logging.getLogger(__name__)
        state = torch.load(restore_bin, map_location="cpu")
        if isinstance(state, dict) and "model" in state and not any(torch.is_tensor(v) for v in state.values()):
            state = state["model"]                         # a training checkpoint wrapped as {"model": state_dict, ...}
        # a checkpoint saved from the reference's nn.DataParallel wrapper carries "module." on every key (model_wrapper.py:146)
        state = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state.items()}
        loaded, unknown, mismatch = _load_flexible(self.model, state, strict_first=False, report=log.info)
        log.warning("restore_checkpoint_pretrained(%s): %d tensors loaded, %d names unknown to this model, %d shape mismatches",
                    restore_bin, len(loaded), len(unknown), len(mismatch))
        if not loaded:
            raise RuntimeError("restore_checkpoint_pretrained(%r): no tensor of the checkpoint matches this model (first keys: %s) -- "
                               "refusing to continue from random weights" % (restore_bin, list(state.keys())[:4]))
        self._after_weights_changed()
        return loaded, unknown, mismatch

    def step(self, batch, eval_mode=False):
        if eval_mode:
            with torch.no_grad():
                output_dict = self.model(**batch)
                if output_dict["loss"] is not None:
                    output_dict["loss"] = output_dict["loss"].mean()
                return output_dict
        self.optimizer.zero_grad()
        gas = self.args.get("gradient_accumulation_steps", 1)
        if self.grad_sync is not None:
            # the gradients of a micro-step that is not followed by optimizer.step() are zeroed by the next call (the reference's
            # order, model_wrapper.py:64): all-reducing them would be pure xGMI traffic
            self.grad_sync.begin_step(sync=(self.called_time + 1) % gas == 0)
        output_dict = self.model(**batch)
        loss = output_dict["loss"]
        if loss.dim() > 0:                                  # (one replica: the loss is a scalar already; .mean() would be a launch + two in backward)
            loss = loss.mean()
        if gas > 1:
            loss = loss / gas
        loss.backward()
        if self.grad_sync is not None:
            self.grad_sync.finish_step()
        if (self.called_time + 1) % gas == 0:
            self.optimizer.step()
            self.global_step += 1
        self.called_time += 1
        return output_dict
