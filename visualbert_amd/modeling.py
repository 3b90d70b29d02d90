"""Drop-in model surface of the reference's hot path, backed by the HIP kernels.

Class names, constructor signatures, forward() signatures, output-dict keys and state-dict keys
follow visualbert/pytorch_pretrained_bert/modeling.py (cited per class; paths relative to
/root/reference/visualbert/pytorch_pretrained_bert/), so visualbert/configs and a caller such as
visualbert/models/model.py:213-223,272-288 work unchanged.  What differs is underneath:

  * every op of forward AND backward runs in libvisualbert_hip.so (visualbert_amd/ops.py);
  * parameters live in ONE flat fp32 arena (ParameterArena): q/k/v weights are adjacent so the
    packed QKV GEMM reads them in place, gradients are written by the kernels straight into a
    flat gradient arena (bucketed RCCL all-reduce and the fused BertAdam work on contiguous ranges),
    and a bf16 shadow arena feeds the MFMA GEMMs in bf16 mode;
  * compute dtype is selectable: torch.float32 (strict parity), torch.bfloat16 (throughput) or "bf16x3" (fp32 activations,
    GEMMs as three bf16 MFMA passes over hi / lo split operands: fp32-class logits at several times the fp32 kernels' speed).

The branches BASELINE.json's configs never take run on the same kernels:
image_text_alignment, bypass_transformer, output_attention_weights, the multichoice / vqa_advanced / flickr heads.  The
`confidence` / `position_embeddings_visual` inputs are accepted and ignored like the reference's embeddings do
(modeling.py:1198-1257).  What is left of that row raises NotImplementedError (attention weights under training-mode
dropout) -- never silently approximated.
"""
import copy
import json
import math

import torch
from torch import nn

from . import _lib, ops


# ------------------------------------------------------------------------------------------------
class BertConfig(object):
    """modeling.py:69-153.  This class is configuration plumbing that has to match the reference field for field (JSON
    files, `to_dict` / `from_dict` round trips, attribute names read all over the callers), so its body follows the
    reference's BertConfig closely; that class derives from pytorch-pretrained-BERT -- Copyright 2018 The Google AI Language
    Team Authors and The HuggingFace Inc. team; Copyright (c) 2018 NVIDIA CORPORATION; licensed under the Apache License,
    Version 2.0 (http://www.apache.org/licenses/LICENSE-2.0)."""

    def __init__(self, vocab_size_or_config_json_file, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
                 max_position_embeddings=512, type_vocab_size=2, initializer_range=0.02):
        if isinstance(vocab_size_or_config_json_file, str):
            with open(vocab_size_or_config_json_file, "r", encoding="utf-8") as reader:
                json_config = json.loads(reader.read())
            for key, value in json_config.items():
                self.__dict__[key] = value
        elif isinstance(vocab_size_or_config_json_file, int):
            self.vocab_size = vocab_size_or_config_json_file
            self.hidden_size = hidden_size
            self.num_hidden_layers = num_hidden_layers
            self.num_attention_heads = num_attention_heads
            self.hidden_act = hidden_act
            self.intermediate_size = intermediate_size
            self.hidden_dropout_prob = hidden_dropout_prob
            self.attention_probs_dropout_prob = attention_probs_dropout_prob
            self.max_position_embeddings = max_position_embeddings
            self.type_vocab_size = type_vocab_size
            self.initializer_range = initializer_range
        else:
            raise ValueError("First argument must be either a vocabulary size (int)"
                             "or the path to a pretrained model config file (str)")

    @classmethod
    def from_dict(cls, json_object):
        config = BertConfig(vocab_size_or_config_json_file=-1)
        for key, value in json_object.items():
            config.__dict__[key] = value
        return config

    @classmethod
    def from_json_file(cls, json_file):
        with open(json_file, "r", encoding="utf-8") as reader:
            text = reader.read()
        return cls.from_dict(json.loads(text))

    def __repr__(self):
        return str(self.to_json_string())

    def to_dict(self):
        return copy.deepcopy(self.__dict__)

    def to_json_string(self):
        return json.dumps(self.to_dict(), indent=2, sort_keys=True) + "\n"


def _check_act(config):
    if getattr(config, "hidden_act", "gelu") != "gelu":
        raise NotImplementedError("visualbert_amd: only hidden_act='gelu' (erf GELU, modeling.py:56-61) has a kernel")


_MASK_OFF = torch.tensor(-10000.0)           # (0-dim, host: a scalar operand of the additive attention mask)


# ------------------------------------------------------------------------------------------------
class ParameterArena(object):
    """Flat fp32 storage for a list of parameters (+ gradient arena, bf16 shadow arena, and the device
    tables the fused BertAdam walks).  Offsets are 64-element aligned so every tensor starts on a
    256-byte boundary (16-byte vector loads of the GEMM, whole cache lines for the optimizer)."""

    ALIGN = 64
    CHUNK = 16384

    def __init__(self, named_params, bucket_of=None):
        named_params = list(named_params)
        if not named_params:
            raise ValueError("empty parameter list")
        dev = named_params[0][1].device
        off = 0
        self.names, self.params, self.offsets = [], [], []
        for n, p in named_params:
            self.names.append(n)
            self.params.append(p)
            self.offsets.append(off)
            off += ops.round_up(p.numel(), self.ALIGN)
        self.numel = off
        self.device = dev
        self.data = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.shadow = torch.zeros(off, dtype=torch.bfloat16, device=dev)
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            view = self.data[o:o + n].view(p.shape)
            view.copy_(p.detach().to(torch.float32))
            p.data = view
            p._vb_arena = self
            p._vb_offset = o
            p._vb_grad = self.grad[o:o + n].view(p.shape)
            p.grad = p._vb_grad
            if p.dim() == 2:
                p._vb_shadow = self.shadow[o:o + n].view(p.shape)
                p._vb_shadow_ver = -1
        # buckets for the gradient all-reduce: contiguous [start, end) ranges, in arena order
        self.bucket_of = bucket_of
        self.touched = set()                    # ids of the parameters a backward pass has written since zero_grad()
        self._touched_dev = None                # device fp32 [n_params] image of `touched` (rewritten when the set changes)
        self._touched_key = None
        self.touched_synced = None              # set by DataParallelGradSync: this step's flags combined over the ranks
        self._tables = None
        self._build_transposed()

    # -- W^T shadows (bf16): dgrad dx = dy W then reads both operands K-contiguously ------------------
    def _build_transposed(self):
        ents = []                                   # (param list, src offset, R, C)
        i = 0
        while i < len(self.params):
            n, p, o = self.names[i], self.params[i], self.offsets[i]
            if n.endswith(".attention.self.query.weight") and i + 2 < len(self.params) and \
                    self.names[i + 1].endswith(".key.weight") and self.names[i + 2].endswith(".value.weight") and \
                    self.offsets[i + 1] == o + p.numel() and self.offsets[i + 2] == o + 2 * p.numel():
                ents.append(([p, self.params[i + 1], self.params[i + 2]], o, 3 * p.size(0), p.size(1)))
                i += 3
                continue
            if p.dim() == 2 and (n.endswith("dense.weight") or n.endswith("word_embeddings.weight")):
                ents.append(([p], o, p.size(0), p.size(1)))
            i += 1
        off = 0
        table, tiles = [], []
        self._t_entries = []
        for ti, (plist, so, R, C) in enumerate(ents):
            ld = ops.round_up(R, 64)
            table += [so, R, C, off, ld]
            for tr in range((R + 63) // 64):
                for tc in range((C + 63) // 64):
                    tiles += [ti, tr, tc]
            self._t_entries.append((plist, off, R, C, ld))
            off += C * ld
        self.shadow_t = torch.zeros(max(off, 1), dtype=torch.bfloat16, device=self.device)
        self._t_table = torch.tensor(table, dtype=torch.int64).to(self.device) if table else None
        self._t_tiles = torch.tensor(tiles, dtype=torch.int64).to(self.device) if tiles else None
        self._t_ntiles = len(tiles) // 3
        for plist, o, R, C, ld in self._t_entries:
            view = self.shadow_t[o:o + C * ld].as_strided((C, R), (ld, 1), o)
            for p in plist:
                p._vb_shadow_t_ver = -1
            if len(plist) == 1:
                plist[0]._vb_shadow_t = view
            else:
                plist[0]._vb_packed_shadow_t = view

    def refresh_transposed(self):
        if self._t_table is None:
            return
        _lib.check(_lib.lib().vb_refresh_transposed_shadow(_lib.ptr(self.shadow), _lib.ptr(self.shadow_t),
                                                           _lib.ptr(self._t_table), _lib.ptr(self._t_tiles),
                                                           self._t_ntiles, _lib.stream_ptr()),
                   "vb_refresh_transposed_shadow")
        for plist, _, _, _, _ in self._t_entries:
            for p in plist:
                p._vb_shadow_t_ver = p._vb_shadow_ver

    def range_of(self, names_prefixes):
        lo, hi = None, None
        for n, p, o in zip(self.names, self.params, self.offsets):
            if any(n.startswith(pref) for pref in names_prefixes):
                lo = o if lo is None else min(lo, o)
                e = o + ops.round_up(p.numel(), self.ALIGN)
                hi = e if hi is None else max(hi, e)
        return lo, hi

    def touched_flags(self):
        """fp32 [n_params] on the arena's device: 1 where a backward pass wrote the parameter's gradient since zero_grad().
        Uploaded only when the set differs from the previous step's (it is the same set step after step)."""
        key = frozenset(self.touched)
        if self._touched_dev is None or key != self._touched_key:
            host = torch.tensor([1.0 if id(p) in key else 0.0 for p in self.params], dtype=torch.float32)
            if self._touched_dev is None:
                self._touched_dev = torch.empty(len(self.params), dtype=torch.float32, device=self.device)
            self._touched_dev.copy_(host, non_blocking=False)
            self._touched_key = key
        return self._touched_dev

    def zero_grad(self):
        self.touched.clear()
        g = self.grad
        nbytes = g.numel() * 4
        if g.is_cuda and nbytes % 16 == 0 and g.data_ptr() % 16 == 0:
            _lib.check(_lib.lib().vb_zero(_lib.ptr(g), nbytes, _lib.stream_ptr()), "vb_zero")
        else:
            g.zero_()

    def tables(self, optimise_flags, decay_flags):
        """device int64 tables for vb_bert_adam_step / vb_refresh_bf16_shadow."""
        tens, chunks = [], []
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            flags = (1 if optimise_flags[i] else 0) | (2 if decay_flags[i] else 0)
            tens += [o, p.numel(), o if p.dim() == 2 else -1, flags]
            n = p.numel()
            count = (n + self.CHUNK - 1) // self.CHUNK
            for s in range(0, n, self.CHUNK):
                chunks += [i, o + s, min(self.CHUNK, n - s), count]
        t = torch.tensor(tens, dtype=torch.int64).to(self.device)
        c = torch.tensor(chunks, dtype=torch.int64).to(self.device)
        return t, c, len(self.params), len(chunks) // 4

    def refresh_shadows(self):
        t, c, nt, nc = self.tables([True] * len(self.params), [False] * len(self.params))
        _lib.check(_lib.lib().vb_refresh_bf16_shadow(_lib.ptr(self.data), _lib.ptr(self.shadow), _lib.ptr(c), nc,
                                                     _lib.ptr(t), _lib.stream_ptr()), "vb_refresh_bf16_shadow")
        for p in self.params:
            if p.dim() == 2:
                p._vb_shadow_ver = p._version
        ops.bump_x3_epoch()
        # a raw write into `data` (broadcast, checkpoint restore) does not move p._version, so the W^T shadows cannot rely
        # on version counters either: re-transpose now (one launch)
        self.refresh_transposed()


# ------------------------------------------------------------------------------------------------
class BertLayerNorm(nn.Module):
    """modeling.py:162-175 (TF-style LayerNorm, eps inside the sqrt) -- the class the reference swaps
    for apex's FusedLayerNorm at import time (:158-160)."""

    def __init__(self, hidden_size, eps=1e-12):
        super(BertLayerNorm, self).__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.bias = nn.Parameter(torch.zeros(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        return ops.LayerNormFn.apply(x, None, self.weight, self.bias, self.variance_epsilon, 0.0, 0.0, 0)


def _drop_p(module_dropout, training):
    return float(module_dropout.p) if training else 0.0


class BertSelfAttention(nn.Module):
    """modeling.py:206-261.  query/key/value keep their own nn.Linear (state-dict contract) but
    their storage is packed [3H, H] / [3H] so one GEMM produces Q|K|V."""

    def __init__(self, config):
        super(BertSelfAttention, self).__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention "
                             "heads (%d)" % (config.hidden_size, config.num_attention_heads))
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = int(config.hidden_size / config.num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        if self.attention_head_size != 64:
            raise NotImplementedError("visualbert_amd: attention kernel is built for head size 64 "
                                      "(BERT-base 768/12, BASELINE config-1 128/2); got %d" % self.attention_head_size)
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)
        self.output_attention_weights = getattr(config, "output_attention_weights", False)
        self._sid = 0
        self._packed_w = self._packed_b = None
        self._packed_shadow = None
        self._packed_shadow_ver = -1

    # -- packed views --------------------------------------------------------------------------
    def _adjacent(self):
        ws = (self.query.weight, self.key.weight, self.value.weight)
        bs = (self.query.bias, self.key.bias, self.value.bias)
        n = ws[0].numel()
        ok = all(w.is_contiguous() and w.untyped_storage().data_ptr() == ws[0].untyped_storage().data_ptr()
                 for w in ws) and all(ws[i].storage_offset() == ws[0].storage_offset() + i * n for i in range(3))
        nb = bs[0].numel()
        ok = ok and all(b.untyped_storage().data_ptr() == bs[0].untyped_storage().data_ptr() for b in bs) and \
            all(bs[i].storage_offset() == bs[0].storage_offset() + i * nb for i in range(3))
        return ok

    def _pack(self):
        """(re)establish adjacency of q/k/v storage when the parameters are not arena-managed."""
        key = (self.query.weight.data_ptr(), self.key.weight.data_ptr(), self.value.weight.data_ptr(),
               self.query.bias.data_ptr())
        if getattr(self, "_pack_key", None) == key:
            return                                        # storage unchanged since the last check
        if self._adjacent():
            self._pack_key = key
            return
        if getattr(self.query.weight, "_vb_arena", None) is not None:
            raise RuntimeError("visualbert_amd: q/k/v parameters are arena-managed but not adjacent")
        H = self.all_head_size
        w = torch.cat([self.query.weight.detach(), self.key.weight.detach(), self.value.weight.detach()], 0).contiguous()
        b = torch.cat([self.query.bias.detach(), self.key.bias.detach(), self.value.bias.detach()], 0).contiguous()
        for i, lin in enumerate((self.query, self.key, self.value)):
            lin.weight.data = w[i * H:(i + 1) * H]
            lin.bias.data = b[i * H:(i + 1) * H]

    @property
    def qkv_weight(self):
        self._pack()
        q = self.query.weight
        t = q.detach().as_strided((3 * q.size(0), q.size(1)), (q.size(1), 1), q.storage_offset())
        return _PackedWeight(t, self)

    @property
    def qkv_bias(self):
        self._pack()
        b = self.query.bias
        return b.detach().as_strided((3 * b.size(0),), (1,), b.storage_offset())

    def qkv_grad_targets(self):
        """(packed dW [3H,H], packed db [3H], direct?)"""
        q = self.query.weight
        gq = getattr(q, "_vb_grad", None)
        if gq is not None and all(getattr(p, "_vb_grad", None) is not None for p in
                                  (self.key.weight, self.value.weight, self.query.bias, self.key.bias, self.value.bias)):
            H = q.size(0)
            gw = gq.as_strided((3 * H, q.size(1)), (q.size(1), 1), gq.storage_offset())
            gb0 = self.query.bias._vb_grad
            gb = gb0.as_strided((3 * H,), (1,), gb0.storage_offset())
            q._vb_arena.touched.update(id(p) for p in (q, self.key.weight, self.value.weight, self.query.bias,
                                                       self.key.bias, self.value.bias))
            return gw, gb, True
        H = q.size(0)
        return (torch.zeros((3 * H, q.size(1)), dtype=torch.float32, device=q.device),
                torch.zeros(3 * H, dtype=torch.float32, device=q.device), False)

    def transpose_for_scores(self, x):
        new_x_shape = x.size()[:-1] + (self.num_attention_heads, self.attention_head_size)
        x = x.view(*new_x_shape)
        return x.permute(0, 2, 1, 3)

    def forward(self, hidden_states, attention_mask):
        if self.output_attention_weights:
            raise NotImplementedError("output_attention_weights: the fused kernel never materialises the probabilities")
        B, S, H = hidden_states.shape
        qkv = _PackedLinearFn.apply(hidden_states, self, self.query.weight, self.query.bias, self.key.weight,
                                    self.key.bias, self.value.weight, self.value.bias)
        mask_add = attention_mask.reshape(B, S).to(torch.float32).contiguous()
        return ops.SelfAttentionCoreFn.apply(qkv, mask_add, self.num_attention_heads,
                                             _drop_p(self.dropout, self.training), self._sid)


class _PackedWeight(object):
    """Duck-typed stand-in handed to ops.weight_for(): the packed [3H,H] alias of q/k/v weights with a
    version number that moves when any of the three parameters is modified through torch."""

    def __init__(self, tensor, owner):
        self._t = tensor
        self._o = owner
        self.device = tensor.device
        self.shape = tensor.shape

    def detach(self):
        return self._t

    @property
    def _version(self):
        o = self._o
        return o.query.weight._version + o.key.weight._version + o.value.weight._version

    @property
    def _vb_shadow(self):
        o = self._o
        sh = getattr(o.query.weight, "_vb_shadow", None)
        if sh is not None and getattr(o.key.weight, "_vb_shadow", None) is not None:
            q = o.query.weight
            return sh.as_strided((3 * q.size(0), q.size(1)), (q.size(1), 1), sh.storage_offset())
        return o._packed_shadow

    @_vb_shadow.setter
    def _vb_shadow(self, v):
        self._o._packed_shadow = v

    @property
    def _vb_shadow_ver(self):
        o = self._o
        if getattr(o.query.weight, "_vb_shadow", None) is not None:
            vs = [o.query.weight._vb_shadow_ver == o.query.weight._version,
                  o.key.weight._vb_shadow_ver == o.key.weight._version,
                  o.value.weight._vb_shadow_ver == o.value.weight._version]
            return self._version if all(vs) else -1
        return o._packed_shadow_ver

    @_vb_shadow_ver.setter
    def _vb_shadow_ver(self, v):
        o = self._o
        if getattr(o.query.weight, "_vb_shadow", None) is not None:
            for p in (o.query.weight, o.key.weight, o.value.weight):
                p._vb_shadow_ver = p._version
        else:
            o._packed_shadow_ver = v

    # W^T [H, 3H] (arena-managed bf16 only)
    @property
    def _vb_shadow_t(self):
        return getattr(self._o.query.weight, "_vb_packed_shadow_t", None)

    @property
    def _vb_shadow_t_ver(self):
        o = self._o
        ok = all(getattr(p, "_vb_shadow_t_ver", -1) == p._version for p in (o.query.weight, o.key.weight, o.value.weight))
        return self._version if ok else -1

    @property
    def _vb_arena(self):
        return getattr(self._o.query.weight, "_vb_arena", None)


@ops.x3_aware
class _PackedLinearFn(torch.autograd.Function):
    """the three Linears of modeling.py:232-234 as one GEMM (stand-alone BertSelfAttention path)."""

    @staticmethod
    def forward(ctx, x, sa, *params):
        B, S, H = x.shape
        x2 = x.reshape(B * S, H)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        w = ops.weight_for(sa.qkv_weight, x2.dtype)
        y = ops.linear_fwd(x2, w, sa.qkv_bias)
        ctx.sa = sa
        ctx.save_for_backward(x2)
        ctx.shape = (B, S, H)
        return y.view(B, S, 3 * H)

    @staticmethod
    def backward(ctx, dy):
        (x2,) = ctx.saved_tensors
        B, S, H = ctx.shape
        sa = ctx.sa
        dy2 = dy.reshape(B * S, 3 * H)
        if dy2.dtype != x2.dtype:
            dy2 = dy2.to(x2.dtype)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        gw, gb, direct = sa.qkv_grad_targets()
        ops.colsum(dy2, gb)
        ops.linear_wgrad(dy2, x2, gw)
        dx = ops.linear_dgrad(dy2, ops.weight_for(sa.qkv_weight, x2.dtype))
        if direct:
            g = [None] * 6
        else:
            g3, b3 = gw.view(3, H, H), gb.view(3, H)
            g = [g3[0], b3[0], g3[1], b3[1], g3[2], b3[2]]
        return (dx.view(B, S, H), None, *g)


class BertSelfOutput(nn.Module):
    """modeling.py:263-274."""

    def __init__(self, config):
        super(BertSelfOutput, self).__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self._sid = 0

    def forward(self, hidden_states, input_tensor):
        h = ops.LinearFn.apply(hidden_states, self.dense.weight, self.dense.bias, None, False)
        return ops.LayerNormFn.apply(h, input_tensor, self.LayerNorm.weight, self.LayerNorm.bias,
                                     self.LayerNorm.variance_epsilon, _drop_p(self.dropout, self.training), 0.0,
                                     self._sid)


class BertAttention(nn.Module):
    """modeling.py:276-293; forward is one fused autograd node (ops.AttentionBlockFn)."""

    def __init__(self, config):
        super(BertAttention, self).__init__()
        self.self = BertSelfAttention(config)
        self.output = BertSelfOutput(config)
        self.output_attention_weights = getattr(config, "output_attention_weights", False)
        self._sid = 0

    def forward(self, input_tensor, attention_mask):
        if self.output_attention_weights:
            raise NotImplementedError("output_attention_weights")
        B, S, H = input_tensor.shape
        mask_add = attention_mask.reshape(B, S).to(torch.float32).contiguous()
        sa, so = self.self, self.output
        return ops.AttentionBlockFn.apply(
            input_tensor, mask_add, self, _drop_p(so.dropout, self.training), _drop_p(sa.dropout, self.training),
            self._sid, sa.query.weight, sa.query.bias, sa.key.weight, sa.key.bias, sa.value.weight, sa.value.bias,
            so.dense.weight, so.dense.bias, so.LayerNorm.weight, so.LayerNorm.bias)


class BertIntermediate(nn.Module):
    """modeling.py:296-305."""

    def __init__(self, config):
        super(BertIntermediate, self).__init__()
        _check_act(config)
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)

    def forward(self, hidden_states):
        return ops.LinearFn.apply(hidden_states, self.dense.weight, self.dense.bias, "gelu", False)


class BertOutput(nn.Module):
    """modeling.py:308-319."""

    def __init__(self, config):
        super(BertOutput, self).__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self._sid = 0

    def forward(self, hidden_states, input_tensor):
        h = ops.LinearFn.apply(hidden_states, self.dense.weight, self.dense.bias, None, False)
        return ops.LayerNormFn.apply(h, input_tensor, self.LayerNorm.weight, self.LayerNorm.bias,
                                     self.LayerNorm.variance_epsilon, _drop_p(self.dropout, self.training), 0.0,
                                     self._sid)


class BertLayer(nn.Module):
    """modeling.py:322-341; two fused autograd nodes per layer (attention block, FFN block).
    `grad_ready_hook`, when set by the data-parallel wrapper, fires once this layer's parameter
    gradients have been enqueued (visualbert_amd/parallel.py overlaps the bucket's all-reduce)."""

    def __init__(self, config):
        super(BertLayer, self).__init__()
        self.attention = BertAttention(config)
        self.intermediate = BertIntermediate(config)
        self.output = BertOutput(config)
        self.output_attention_weights = getattr(config, "output_attention_weights", False)
        self.layer_index = 0
        self.grad_ready_hook = None

    def set_index(self, i):
        self.layer_index = i
        base = 16 + 8 * i                       # dropout stream ids, unique per site
        self.attention._sid = base
        self.attention.self._sid = base
        self.attention.output._sid = base + 1
        self.output._sid = base + 4

    def forward(self, hidden_states, attention_mask):
        if self.output_attention_weights:                   # modeling.py:331-336: (layer_output, attention_probs)
            if self.training and self.attention.self.dropout.p > 0.0:
                raise NotImplementedError("output_attention_weights in training mode would return the probabilities "
                                          "after dropout (modeling.py:251); only eval / p = 0 is provided")
            mask_add = attention_mask.reshape(hidden_states.size(0), hidden_states.size(1)).to(torch.float32).contiguous()
            with torch.no_grad():
                probs = ops.attention_probs(hidden_states, self.attention.self, mask_add)
            return self._forward_fused(hidden_states, attention_mask), probs
        return self._forward_fused(hidden_states, attention_mask)

    def _forward_fused(self, hidden_states, attention_mask):
        if self.grad_ready_hook is not None and torch.is_grad_enabled() and hidden_states.requires_grad:
            hidden_states = _GradReadyFn.apply(hidden_states, self)
        (at, sa, so, im, om), _, params = ops.layer_params(self)     # the sub-modules and their 16 parameters, cached on the layer
        B, S, H = hidden_states.shape
        mask_add = attention_mask.reshape(B, S)
        if mask_add.dtype != torch.float32 or not mask_add.is_contiguous():
            mask_add = mask_add.to(torch.float32).contiguous()
        # parameter order: q.w q.b k.w k.b v.w v.b | self-output dense.w dense.b LN.w LN.b | intermediate dense.w dense.b |
        # output dense.w dense.b LN.w LN.b
        return ops.BertLayerFn.apply(
            hidden_states, mask_add, self, _drop_p(om.dropout, self.training), _drop_p(sa.dropout, self.training), *params)

    def forward_unfused(self, hidden_states, attention_mask):
        """the same layer as two autograd nodes (attention block, FFN block) -- kept for tests that compare
        the single-call path against the op-by-op path."""
        attention_output = self.attention(hidden_states, attention_mask)
        im, om = self.intermediate, self.output
        return ops.FFNBlockFn.apply(attention_output, im, om, _drop_p(om.dropout, self.training), om._sid,
                                    im.dense.weight, im.dense.bias, om.dense.weight, om.dense.bias,
                                    om.LayerNorm.weight, om.LayerNorm.bias)


class _GradReadyFn(torch.autograd.Function):
    """identity whose backward runs after everything of the layer above it in the graph: by then the
    layer's wgrad kernels are enqueued, so its gradient bucket can be handed to RCCL."""

    @staticmethod
    def forward(ctx, x, layer):
        ctx.layer = layer
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        hook = ctx.layer.grad_ready_hook
        if hook is not None:
            hook(ctx.layer.layer_index)
        return g, None


class BertEncoder(nn.Module):
    """modeling.py:344-371."""

    def __init__(self, config):
        super(BertEncoder, self).__init__()
        self.layer = nn.ModuleList([BertLayer(config) for _ in range(config.num_hidden_layers)])
        for i, l in enumerate(self.layer):
            l.set_index(i)
        self.output_attention_weights = getattr(config, "output_attention_weights", False)

    def forward(self, hidden_states, attention_mask, output_all_encoded_layers=True):
        attn_data_list = []
        all_encoder_layers = []
        for layer_module in self.layer:
            if self.output_attention_weights:               # modeling.py:352-362
                hidden_states, attention_weights = layer_module(hidden_states, attention_mask)
                attn_data_list.append(attention_weights)
            else:
                hidden_states = layer_module(hidden_states, attention_mask)
            if output_all_encoded_layers:
                all_encoder_layers.append(hidden_states)
        if not output_all_encoded_layers:
            all_encoder_layers.append(hidden_states)
        if self.output_attention_weights:
            return all_encoder_layers, attn_data_list
        return all_encoder_layers


class BertPooler(nn.Module):
    """modeling.py:374-386: tanh(W h[:, 0] + b); the first-token rows are read in place (lda = S*H)."""

    def __init__(self, config):
        super(BertPooler, self).__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.activation = nn.Tanh()

    def forward(self, hidden_states):
        first_token_tensor = hidden_states[:, 0]
        return ops.LinearFn.apply(first_token_tensor, self.dense.weight, self.dense.bias, "tanh", False)


class BertPredictionHeadTransform(nn.Module):
    """modeling.py:389-401."""

    def __init__(self, config):
        super(BertPredictionHeadTransform, self).__init__()
        _check_act(config)
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=1e-12)

    def forward(self, hidden_states):
        h = ops.LinearFn.apply(hidden_states, self.dense.weight, self.dense.bias, "gelu", False)
        return self.LayerNorm(h)


class BertLMPredictionHead(nn.Module):
    """modeling.py:404-420; decoder weight tied to the word embeddings (:414)."""

    def __init__(self, config, bert_model_embedding_weights):
        super(BertLMPredictionHead, self).__init__()
        self.transform = BertPredictionHeadTransform(config)
        self.decoder = nn.Linear(bert_model_embedding_weights.size(1), bert_model_embedding_weights.size(0), bias=False)
        self.decoder.weight = bert_model_embedding_weights
        self.bias = nn.Parameter(torch.zeros(bert_model_embedding_weights.size(0)))

    def forward(self, hidden_states):
        h = self.transform(hidden_states)
        return ops.LinearFn.apply(h, self.decoder.weight, self.bias, None, True)


class BertPreTrainingHeads(nn.Module):
    """modeling.py:443-453."""

    def __init__(self, config, bert_model_embedding_weights):
        super(BertPreTrainingHeads, self).__init__()
        self.predictions = BertLMPredictionHead(config, bert_model_embedding_weights)
        self.seq_relationship = nn.Linear(config.hidden_size, 2)

    def forward(self, sequence_output, pooled_output):
        prediction_scores = self.predictions(sequence_output)
        rel, _ = ops.SmallLinearCEFn.apply(pooled_output, None, -1, self.seq_relationship.weight,
                                           self.seq_relationship.bias)
        return prediction_scores, rel


# ------------------------------------------------------------------------------------------------
# The architectures behind the model names the reference resolves to archive URLs (modeling.py:44-52,
# PRETRAINED_MODEL_ARCHIVE_MAP).  Each archive's bert_config.json is public knowledge (Google's BERT release); with no network
# the name resolves to that architecture and -- unless a local directory provides pytorch_model.bin -- to init_bert_weights'
# random initialisation, which is what an offline reference run with `random_initialize=True` produces.
def _arch(vocab, hidden=768, layers=12, heads=12, inter=3072):
    return dict(vocab_size=vocab, hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                intermediate_size=inter, hidden_act="gelu", hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1,
                max_position_embeddings=512, type_vocab_size=2, initializer_range=0.02)


PRETRAINED_MODEL_ARCHITECTURES = {
    "bert-base-uncased": _arch(30522),
    "bert-large-uncased": _arch(30522, 1024, 24, 16, 4096),
    "bert-base-cased": _arch(28996),
    "bert-large-cased": _arch(28996, 1024, 24, 16, 4096),
    "bert-base-multilingual-uncased": _arch(105879),
    "bert-base-multilingual-cased": _arch(119547),
    "bert-base-chinese": _arch(21128),
}
CONFIG_NAME = "bert_config.json"
WEIGHTS_NAME = "pytorch_model.bin"


def resolve_pretrained(pretrained_model_name, cache_dir=None):
    """-> (BertConfig, directory holding pytorch_model.bin or None).  The reference (modeling.py:511-544) maps a known name to
    an S3 archive and anything else to a path; here a path (or <cache_dir>/<name>, or $VISUALBERT_AMD_BERT_DIR/<name>) that
    holds bert_config.json wins, then a known name falls back to its built-in architecture."""
    import os
    candidates = [pretrained_model_name]
    for root in (cache_dir, os.environ.get("VISUALBERT_AMD_BERT_DIR")):
        if root:
            candidates.append(os.path.join(str(root), str(pretrained_model_name)))
    for c in candidates:
        if os.path.isfile(os.path.join(str(c), CONFIG_NAME)):
            return BertConfig.from_json_file(os.path.join(str(c), CONFIG_NAME)), str(c)
    if pretrained_model_name in PRETRAINED_MODEL_ARCHITECTURES:
        return BertConfig.from_dict(PRETRAINED_MODEL_ARCHITECTURES[pretrained_model_name]), None
    raise FileNotFoundError(
        "visualbert_amd: model name %r is neither one of %s nor a directory containing %s (no network access: archives are "
        "not downloaded)" % (pretrained_model_name, ", ".join(sorted(PRETRAINED_MODEL_ARCHITECTURES)), CONFIG_NAME))


class PreTrainedBertModel(nn.Module):
    """modeling.py:459-596: init_bert_weights and from_pretrained.  There is no network here, so from_pretrained never
    downloads: a local directory supplies bert_config.json [+ pytorch_model.bin]; a bare model name of the reference's archive
    map resolves to its architecture (resolve_pretrained) with randomly initialised weights and a logged warning."""

    def __init__(self, config, *inputs, **kwargs):
        super(PreTrainedBertModel, self).__init__()
        if not isinstance(config, BertConfig):
            raise ValueError("Parameter config in `{}(config)` should be an instance of class `BertConfig`.".format(
                self.__class__.__name__))
        self.config = config

    def init_bert_weights(self, module):
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.config.initializer_range)
        elif isinstance(module, BertLayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()

    @classmethod
    def from_pretrained(cls, pretrained_model_name, state_dict=None, cache_dir=None, random_initialize=False,
                        *inputs, **kwargs):
        import logging
        import os
        config, directory = resolve_pretrained(pretrained_model_name, cache_dir)
        model = cls(config, *inputs, **kwargs)
        if random_initialize:
            return model
        if state_dict is None:
            weights = os.path.join(directory, WEIGHTS_NAME) if directory else None
            if weights is None or not os.path.isfile(weights):
                logging.getLogger(__name__).warning(
                    "visualbert_amd: no %s for %r on this machine (archives are not downloaded): the %s keeps its random "
                    "initialisation; load weights with ModelWrapper.restore_checkpoint_pretrained / `restore_bin`",
                    WEIGHTS_NAME, pretrained_model_name, cls.__name__)
                return model
            state_dict = torch.load(weights, map_location="cpu")
        renamed = {}
        for key, value in state_dict.items():      # legacy gamma/beta names, modeling.py:556-568
            nk = key.replace("gamma", "weight").replace("beta", "bias")
            renamed[nk] = value
        own = model.state_dict()
        prefix = "" if hasattr(model, "bert") else "bert."
        with torch.no_grad():
            for k, v in renamed.items():
                kk = k[len(prefix):] if prefix and k.startswith(prefix) else k
                if kk in own and own[kk].shape == v.shape:
                    own[kk].copy_(v)
        return model


class BertEmbeddingsWithVisualEmbedding(nn.Module):
    """modeling.py:1168-1257."""

    def __init__(self, config):
        super(BertEmbeddingsWithVisualEmbedding, self).__init__()
        self.word_embeddings = nn.Embedding(config.vocab_size, config.hidden_size)
        self.position_embeddings = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.token_type_embeddings = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self.token_type_embeddings_visual = nn.Embedding(config.type_vocab_size, config.hidden_size)
        self.position_embeddings_visual = nn.Embedding(config.max_position_embeddings, config.hidden_size)
        self.projection = nn.Linear(config.visual_embedding_dim, config.hidden_size)
        self.compute_dtype = torch.float32

    def special_intialize(self, method_type=0):
        """modeling.py:1191-1196 (sic): copy the text tables into the *_visual tables (in place, so the
        parameters stay inside the arena)."""
        with torch.no_grad():
            self.token_type_embeddings_visual.weight.copy_(self.token_type_embeddings.weight)
            self.position_embeddings_visual.weight.copy_(self.position_embeddings.weight)

    def forward(self, input_ids, token_type_ids=None, visual_embeddings=None, visual_embeddings_type=None,
                position_embeddings_visual=None, image_text_alignment=None, confidence=None):
        if visual_embeddings is None:
            image_text_alignment = None
        if visual_embeddings is not None and visual_embeddings_type is None:
            visual_embeddings_type = torch.zeros(visual_embeddings.shape[:2], dtype=torch.long,
                                                 device=input_ids.device)
        m = self
        return ops.EmbeddingsFn.apply(
            m, input_ids, token_type_ids, visual_embeddings, visual_embeddings_type, image_text_alignment,
            self.compute_dtype, _drop_p(self.dropout, self.training), 8,
            m.word_embeddings.weight, m.position_embeddings.weight, m.token_type_embeddings.weight,
            m.LayerNorm.weight, m.LayerNorm.bias, m.token_type_embeddings_visual.weight,
            m.position_embeddings_visual.weight, m.projection.weight, m.projection.bias)


class BertVisualModel(PreTrainedBertModel):
    """modeling.py:1260-1333."""

    def __init__(self, config):
        super(BertVisualModel, self).__init__(config)
        self.embeddings = BertEmbeddingsWithVisualEmbedding(config)
        self.encoder = BertEncoder(config)
        self.pooler = BertPooler(config)
        self.bypass_transformer = getattr(config, "bypass_transformer", False)
        if self.bypass_transformer:
            self.additional_layer = BertLayer(config)           # modeling.py:1268-1269
            self.additional_layer.set_index(config.num_hidden_layers)
        self.output_attention_weights = getattr(config, "output_attention_weights", False)
        if self.bypass_transformer and self.output_attention_weights:
            raise NotImplementedError("bypass_transformer with output_attention_weights: the reference's bypass branch "
                                      "indexes the encoder's (layers, weights) tuple as a layer list (modeling.py:1306-1312)")
        self.apply(self.init_bert_weights)

    def forward(self, input_ids, token_type_ids, attention_mask, visual_embeddings, position_embeddings_visual,
                visual_embeddings_type, image_text_alignment, confidence, output_all_encoded_layers=True):
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        if token_type_ids is None:
            token_type_ids = torch.zeros_like(input_ids)
        extended_attention_mask = attention_mask.unsqueeze(1).unsqueeze(2)
        if attention_mask.is_floating_point():
            # the reference's own three operations (modeling.py:1293-1294): a fractional mask rounds as it does there
            extended_attention_mask = (1.0 - extended_attention_mask.to(dtype=torch.float32)) * -10000.0
        else:
            # integer / bool masks (what every data path of the reference produces): (1.0 - mask) * -10000.0 in fp32 as ONE launch,
            # -10000 + 10000 * mask with the constant a 0-dim host tensor -- the same values (0 or -10000) exactly
            extended_attention_mask = torch.add(_MASK_OFF, extended_attention_mask, alpha=10000.0)
        embedding_output = self.embeddings(input_ids, token_type_ids, visual_embeddings=visual_embeddings,
                                           position_embeddings_visual=position_embeddings_visual,
                                           visual_embeddings_type=visual_embeddings_type,
                                           image_text_alignment=image_text_alignment, confidence=confidence)
        if self.bypass_transformer and visual_embeddings is not None:
            # modeling.py:1299-1314: the encoder sees the text only; one more BertLayer sees text + regions
            assert not output_all_encoded_layers
            text_length = input_ids.size(1)
            text_embedding_output = embedding_output[:, :text_length, :].contiguous()
            visual_part = embedding_output[:, text_length:, :]
            text_extended_attention_mask = extended_attention_mask[:, :, :text_length, :text_length]
            encoded_layers = self.encoder(text_embedding_output, text_extended_attention_mask,
                                          output_all_encoded_layers=output_all_encoded_layers)
            sequence_output = encoded_layers[-1]
            new_input = torch.cat((sequence_output, visual_part), dim=1)
            final_sequence_output = self.additional_layer(new_input, extended_attention_mask)
            pooled_output = self.pooler(final_sequence_output)
            return final_sequence_output, pooled_output
        attn_data_list = None
        if self.output_attention_weights:                   # modeling.py:1316-1324
            encoded_layers, attn_data_list = self.encoder(embedding_output, extended_attention_mask,
                                                          output_all_encoded_layers=output_all_encoded_layers)
        else:
            encoded_layers = self.encoder(embedding_output, extended_attention_mask,
                                          output_all_encoded_layers=output_all_encoded_layers)
        sequence_output = encoded_layers[-1]
        pooled_output = self.pooler(sequence_output)
        if not output_all_encoded_layers:
            encoded_layers = encoded_layers[-1]
        if self.output_attention_weights:
            return encoded_layers, pooled_output, attn_data_list
        return encoded_layers, pooled_output


def transform_to_batch_sequence(tensor):
    """modeling.py:1675-1683."""
    if tensor is not None:
        if len(tensor.size()) == 2:
            return tensor
        assert len(tensor.size()) == 3
        return tensor.contiguous().view(-1, tensor.size(-1))
    return None


def transform_to_batch_sequence_dim(tensor):
    """modeling.py:1685-1693."""
    if tensor is not None:
        if len(tensor.size()) == 3:
            return tensor
        assert len(tensor.size()) == 4
        return tensor.contiguous().view(-1, tensor.size(-2), tensor.size(-1))
    return None


class TrainVisualBERTObjective(PreTrainedBertModel):
    """modeling.py:1335-1598 -- same constructor / forward signature / output dict.  Extra keyword
    `compute_dtype` selects fp32 (parity) or bf16 (throughput) kernels; `.half()` maps to bf16."""

    SUPPORTED_HEADS = ("pretraining", "vqa", "nlvr", "multichoice", "vqa_advanced", "flickr")

    def __init__(self, config, training_head_type, visual_embedding_dim=512, hard_cap_seq_len=None, cut_first="text",
                 embedding_strategy="plain", bypass_transformer=False, output_attention_weights=False,
                 compute_dtype=torch.float32):
        super(TrainVisualBERTObjective, self).__init__(config)
        config.visual_embedding_dim = visual_embedding_dim
        config.embedding_strategy = embedding_strategy
        config.bypass_transformer = bypass_transformer
        config.output_attention_weights = output_attention_weights
        self.output_attention_weights = output_attention_weights
        self.cut_first = cut_first
        self.hard_cap_seq_len = hard_cap_seq_len
        self.bert = BertVisualModel(config)
        self.training_head_type = training_head_type
        self._zero_types = {}
        self.sparse_mlm_head = False       # opt-in: MLM head over the labelled positions only (changes `logits` to [n, V])
        if training_head_type == "pretraining":
            self.cls = BertPreTrainingHeads(config, self.bert.embeddings.word_embeddings.weight)
        elif training_head_type == "multichoice":           # modeling.py:1353-1356
            self.dropout = nn.Dropout(config.hidden_dropout_prob)
            self.classifier = nn.Linear(config.hidden_size, 1)
            self.num_choices = 4                            # For VCR
        elif training_head_type == "vqa_advanced":          # modeling.py:1361-1362
            self.cls = BertPreTrainingHeads(config, self.bert.embeddings.word_embeddings.weight)
        elif training_head_type == "flickr":                # modeling.py:1366-1369
            self.dropout = nn.Dropout(config.hidden_dropout_prob)
            self.cls = BertPreTrainingHeads(config, self.bert.embeddings.word_embeddings.weight)
            self.flickr_attention = FlickrAttention(config)
        elif training_head_type == "vqa":
            self.dropout = nn.Dropout(config.hidden_dropout_prob)
            self.classifier = nn.Linear(config.hidden_size, 3129)
        elif training_head_type == "nlvr":
            self.dropout = nn.Dropout(config.hidden_dropout_prob)
            self.classifier = nn.Linear(config.hidden_size, 2)
        else:
            raise NotImplementedError("training_head_type %r: the reference has %s"
                                      % (training_head_type, ", ".join(self.SUPPORTED_HEADS)))
        self.apply(self.init_bert_weights)
        self.arena = None
        self.set_compute_dtype(compute_dtype)
        self.build_arena()

    # -- storage ---------------------------------------------------------------------------------
    def set_compute_dtype(self, dtype):
        """torch.float32: every kernel in fp32 (fp32-input MFMA; the strict parity mode).  torch.bfloat16 (also what .half()
        selects): bf16 storage and MFMA operands, fp32 accumulation.  "bf16x3": fp32 activations and fp32-class results
        with the GEMMs on the bf16 matrix pipe -- operands split into hi + lo bf16 planes, three passes (VB_BF16X3)."""
        x3 = isinstance(dtype, str) and dtype.lower() == "bf16x3"
        if x3:
            dtype = torch.float32
        if dtype in (torch.float16, torch.half):
            dtype = torch.bfloat16              # gfx950 path: bf16 storage + fp32 accumulate
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute dtype must be float32, bfloat16 or 'bf16x3'")
        self.compute_dtype = dtype
        self.gemm_x3 = x3
        self.bert.embeddings.compute_dtype = dtype
        return self

    def half(self):
        """the reference calls model.half() under fp16 (models/model_wrapper.py:143-145); here the
        masters stay fp32 and compute switches to bf16."""
        return self.set_compute_dtype(torch.bfloat16)

    def bfloat16(self):
        return self.set_compute_dtype(torch.bfloat16)

    def float(self):
        return self.set_compute_dtype(torch.float32)

    def _arena_order(self):
        """arena order = reverse of the order gradients become final in backward is NOT needed; what
        matters is that each all-reduce bucket is contiguous: [embeddings | layer 0 | ... | layer L-1 |
        pooler + heads]."""
        seen, out = set(), []
        for n, p in self.named_parameters():
            if id(p) in seen:
                continue
            seen.add(id(p))
            out.append((n, p))
        # q/k/v weights (and biases) of every BertSelfAttention must be adjacent: the packed QKV GEMM
        # reads them in place as one [3H, H] matrix
        rank = {".query.weight": 0, ".key.weight": 1, ".value.weight": 2,
                ".query.bias": 3, ".key.bias": 4, ".value.bias": 5}
        ordered, i = [], 0
        while i < len(out):
            n = out[i][0]
            if ".attention.self." in n:
                j = i
                while j < len(out) and ".attention.self." in out[j][0] and \
                        out[j][0].rsplit(".attention.self.", 1)[0] == n.rsplit(".attention.self.", 1)[0]:
                    j += 1
                grp = sorted(out[i:j], key=lambda t: min([v for k, v in rank.items() if t[0].endswith(k)] + [9]))
                ordered += grp
                i = j
            else:
                ordered.append(out[i])
                i += 1
        return ordered

    def build_arena(self):
        self.arena = ParameterArena(self._arena_order())
        return self.arena

    def _apply(self, fn, recurse=True):
        out = super(TrainVisualBERTObjective, self)._apply(fn, recurse)
        if getattr(self, "arena", None) is not None or hasattr(self, "training_head_type"):
            for p in self.parameters():
                if p.dtype != torch.float32:
                    p.data = p.data.to(torch.float32)
            self.build_arena()
        return out

    def zero_grad(self, set_to_none=False):
        """nn.Module.zero_grad() would set every p.grad to None and leave the flat gradient arena -- which the kernels
        accumulate into -- untouched, so a loop calling model.zero_grad() instead of optimizer.zero_grad() would sum
        gradients across steps.  One memset of the arena; p.grad stays bound to its arena view."""
        if self.arena is not None:
            self.arena.zero_grad()
            for p in self.arena.params:
                p.grad = p._vb_grad

    def _check_inputs(self, input_ids, token_type_ids, masked_lm_labels):
        """the reference fails loudly on corrupt inputs (nn.Embedding raises on an out-of-range index, CrossEntropyLoss on
        a label >= V); the gather kernels clamp instead of faulting, so the range checks live here.  The sequence-length
        check is free and always on; the value-range checks read the tensors back (a device synchronisation per
        forward), so they are the debug mode VB_CHECK_INPUTS=1."""
        cfg = self.config
        T = input_ids.size(-1)
        if T > cfg.max_position_embeddings:
            raise IndexError("sequence length %d exceeds max_position_embeddings %d" % (T, cfg.max_position_embeddings))
        import os
        if os.environ.get("VB_CHECK_INPUTS", "0") != "1":
            return
        lo, hi = int(input_ids.min()), int(input_ids.max())
        if lo < 0 or hi >= cfg.vocab_size:
            raise IndexError("input_ids out of range [0, %d): min %d max %d" % (cfg.vocab_size, lo, hi))
        if token_type_ids is not None and token_type_ids.numel():
            lo, hi = int(token_type_ids.min()), int(token_type_ids.max())
            if lo < 0 or hi >= cfg.type_vocab_size:
                raise IndexError("token_type_ids out of range [0, %d): min %d max %d" % (cfg.type_vocab_size, lo, hi))
        if masked_lm_labels is not None and masked_lm_labels.numel():
            lo, hi = int(masked_lm_labels.min()), int(masked_lm_labels.max())
            if lo < -1 or hi >= cfg.vocab_size:
                raise IndexError("masked_lm_labels out of range [-1, %d): min %d max %d" % (cfg.vocab_size, lo, hi))

    def bucket_ranges(self):
        """[(start, end)] element ranges of the gradient arena, one per all-reduce bucket, listed in the
        order backward completes them: heads+pooler, layer L-1 ... layer 0, embeddings."""
        a = self.arena
        L = len(self.bert.encoder.layer)
        r = []
        head_pref = ["bert.pooler.", "bert.additional_layer.", "cls.", "classifier.", "flickr_attention."]
        lo, hi = a.range_of(head_pref)
        r.append(("heads", lo, hi))
        for i in reversed(range(L)):
            lo, hi = a.range_of(["bert.encoder.layer.%d." % i])
            r.append(("layer%d" % i, lo, hi))
        lo, hi = a.range_of(["bert.embeddings."])
        r.append(("embeddings", lo, hi))
        return r

    # -- forward ---------------------------------------------------------------------------------
    def forward(self, *args, **kwargs):
        with ops.x3_scope(getattr(self, "gemm_x3", False)):          # GEMM mode of THIS model, for forward and (via ctx) backward
            return self._forward(*args, **kwargs)

    def _forward(self, input_ids, token_type_ids, input_mask, visual_embeddings, position_embeddings_visual, image_mask,
                 image_text_alignment=None, confidence=None, visual_embeddings_type=None, label=None,
                 flickr_position=None, masked_lm_labels=None, image_lm_lables=None, is_random_next=None,
                 output_all_encoded_layers=False):
        # `confidence` and `position_embeddings_visual` are accepted and IGNORED, exactly as the reference does: its
        # embeddings take both arguments and never read them (modeling.py:1198-1257; :1383, :1403 pass them through)
        flat_input_ids = transform_to_batch_sequence(input_ids)
        flat_token_type_ids = transform_to_batch_sequence(token_type_ids)
        self._check_inputs(flat_input_ids, flat_token_type_ids, masked_lm_labels)
        flat_input_mask = transform_to_batch_sequence(input_mask)
        flat_image_mask = transform_to_batch_sequence(image_mask)
        flat_masked_lm_labels = transform_to_batch_sequence(masked_lm_labels)
        flat_visual_embeddings = transform_to_batch_sequence_dim(visual_embeddings)
        flat_image_text_alignment = transform_to_batch_sequence_dim(image_text_alignment)
        if visual_embeddings_type is not None:
            visual_embeddings_type = transform_to_batch_sequence(visual_embeddings_type)
        elif flat_image_mask is not None:
            key = (tuple(flat_image_mask.shape), flat_image_mask.device)
            visual_embeddings_type = self._zero_types.get(key)            # all-zero type ids, read-only: made once per shape
            if visual_embeddings_type is None:
                if len(self._zero_types) > 8:
                    self._zero_types.clear()
                visual_embeddings_type = self._zero_types[key] = torch.zeros_like(flat_image_mask, dtype=torch.long)

        if flat_image_mask is not None:
            assert image_lm_lables is None
            if flat_masked_lm_labels is not None:
                assert flat_masked_lm_labels.size(-1) == flat_input_mask.size(-1)
            # integer path in one kernel (bit-exact): attention_mask = cat(input_mask, image_mask),
            # LM labels extended with -1 over the visual slots (modeling.py:1417-1426)
            flat_attention_mask, _, ext = ops.prepare_inputs(flat_input_mask, None, flat_image_mask.contiguous(),
                                                             flat_masked_lm_labels, flat_image_mask.size(1))
            if flat_masked_lm_labels is not None:
                flat_masked_lm_labels = ext
        else:
            flat_attention_mask = flat_input_mask
        if self.training_head_type in ("pretraining", "vqa_advanced") and flat_masked_lm_labels is not None \
                and not output_all_encoded_layers:
            flat_masked_lm_labels = flat_masked_lm_labels.contiguous()
            ops.plan_masked_rows(flat_masked_lm_labels)      # count the labelled rows now, read the count 12 layers later

        if self.output_attention_weights:                    # modeling.py:1428-1442
            _, _, attention_weights = self.bert(
                flat_input_ids, flat_token_type_ids, flat_attention_mask, visual_embeddings=flat_visual_embeddings,
                position_embeddings_visual=None, visual_embeddings_type=visual_embeddings_type,
                image_text_alignment=flat_image_text_alignment, confidence=None,
                output_all_encoded_layers=output_all_encoded_layers)
            return {"attention_weights": attention_weights, "loss": None}

        sequence_output, pooled_output = self.bert(
            flat_input_ids, flat_token_type_ids, flat_attention_mask, visual_embeddings=flat_visual_embeddings,
            position_embeddings_visual=None, visual_embeddings_type=visual_embeddings_type,
            image_text_alignment=flat_image_text_alignment, confidence=None,
            output_all_encoded_layers=output_all_encoded_layers)

        output_dict = {}
        if output_all_encoded_layers:
            output_dict["sequence_output"] = sequence_output
            output_dict["pooled_output"] = pooled_output
            output_dict["loss"] = None
            return output_dict

        if self.training_head_type == "pretraining":
            pred = self.cls.predictions
            tr = pred.transform
            if self.sparse_mlm_head and flat_masked_lm_labels is not None:
                # opt-in: the head runs over the labelled positions only; `logits` is [n, V] and
                # `logits_rows` says which of the B*S positions they are -- the reference returns [B, S, V]
                logits, logit_rows, mlm_loss = ops.SparseMLMHeadLossFn.apply(
                    sequence_output, flat_masked_lm_labels, pred, pred.decoder.weight, pred.bias, tr.dense.weight,
                    tr.dense.bias, tr.LayerNorm.weight, tr.LayerNorm.bias)
                output_dict["logits_rows"] = logit_rows
            else:
                logits, mlm_loss = ops.MLMHeadLossFn.apply(
                    sequence_output, flat_masked_lm_labels, pred, pred.decoder.weight, pred.bias, tr.dense.weight,
                    tr.dense.bias, tr.LayerNorm.weight, tr.LayerNorm.bias)
            rel, nsp_loss = ops.SmallLinearCEFn.apply(pooled_output, is_random_next, -1,
                                                      self.cls.seq_relationship.weight, self.cls.seq_relationship.bias)
            output_dict["logits"] = logits
            output_dict["seq_relationship_score"] = rel
            output_dict["loss"] = None
            if flat_masked_lm_labels is not None and is_random_next is not None:
                output_dict["next_sentence_loss"] = nsp_loss
                output_dict["masked_lm_loss"] = mlm_loss
                output_dict["loss"] = mlm_loss + nsp_loss
            if flat_masked_lm_labels is not None and is_random_next is None:
                output_dict["masked_lm_loss"] = mlm_loss
                output_dict["loss"] = mlm_loss
            return output_dict

        if self.training_head_type == "multichoice":         # modeling.py:1488-1500
            p = _drop_p(self.dropout, self.training)
            po = _small_dropout(pooled_output, p) if p > 0.0 else pooled_output
            logits, loss = ops.SmallLinearCEFn.apply(po, label, -100, self.classifier.weight, self.classifier.bias,
                                                     self.num_choices)
            output_dict["logits"] = logits
            output_dict["loss"] = loss if label is not None else None
            return output_dict

        if self.training_head_type == "vqa_advanced":        # modeling.py:1527-1554
            pred = self.cls.predictions
            tr = pred.transform
            logits, mlm_loss = ops.MLMHeadLossFn.apply(
                sequence_output, flat_masked_lm_labels, pred, pred.decoder.weight, pred.bias, tr.dense.weight,
                tr.dense.bias, tr.LayerNorm.weight, tr.LayerNorm.bias)
            rel, _ = ops.SmallLinearCEFn.apply(pooled_output, None, -1, self.cls.seq_relationship.weight,
                                               self.cls.seq_relationship.bias)
            output_dict["logits"] = logits
            output_dict["seq_relationship_score"] = rel
            output_dict["masked_lm_loss"] = mlm_loss
            output_dict["loss"] = mlm_loss
            # a sample counts when every labelled token is predicted (the reference loops over a numpy copy)
            lab = flat_masked_lm_labels.view(flat_input_ids.size(0), -1)
            hit = ((lab == -1) | (logits.argmax(-1).view(lab.shape) == lab)).all(dim=1)
            output_dict["accuracy"] = float(hit.sum()) / lab.size(0)
            return output_dict

        if self.training_head_type == "flickr":              # modeling.py:1568-1598
            if flickr_position is not None:
                loss, acc, upper, entities_num = ops.FlickrHeadLossFn.apply(
                    sequence_output, flickr_position, flat_image_mask, label, flat_input_mask.size(1),
                    self.flickr_attention.attention_head_size, self.flickr_attention.query.weight,
                    self.flickr_attention.query.bias, self.flickr_attention.key.weight,
                    self.flickr_attention.key.bias)
                output_dict["loss"] = loss
                output_dict["accuracy"] = acc / entities_num
                output_dict["upperbound_accuracy"] = upper / entities_num
                output_dict["entity_num"] = entities_num
            return output_dict

        if self.training_head_type == "vqa":
            logits, loss, acc, _ = ops.VQAHeadLossFn.apply(
                sequence_output, flat_input_mask, label, _drop_p(self.dropout, self.training), 9,
                self.classifier.weight, self.classifier.bias)
            output_dict["logits"] = logits
            output_dict["loss"] = None
            output_dict["accuracy"] = None
            if label is not None:
                output_dict["loss"] = loss
                output_dict["accuracy"] = acc
            return output_dict

        if self.training_head_type == "nlvr":
            p = _drop_p(self.dropout, self.training)
            po = pooled_output
            if p > 0.0:
                po = _small_dropout(po, p)
            logits, loss = ops.SmallLinearCEFn.apply(po, label, -100, self.classifier.weight, self.classifier.bias)
            output_dict["logits"] = logits
            output_dict["loss"] = loss if label is not None else None
            return output_dict
        raise NotImplementedError(self.training_head_type)


class FlickrAttention(nn.Module):
    """modeling.py:1602-1648: ONE attention head of size H / num_attention_heads; only its query and key
    projections are used (no value, no softmax) -- the scores feed the grounding loss."""

    def __init__(self, config):
        super(FlickrAttention, self).__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (config.hidden_size, config.num_attention_heads))
        self.num_attention_heads = 1
        self.attention_head_size = int(config.hidden_size / config.num_attention_heads)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(config.hidden_size, self.all_head_size)
        self.value = nn.Linear(config.hidden_size, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)


def _small_dropout(x, p):
    """nn.Dropout on the [B, H] pooled vector of the multichoice / NLVR2 fine-tune heads (modeling.py:1495, 1557): the
    counter-based HIP dropout of every other site (vb_dropout; mask regenerated in backward, stream id 10)."""
    return ops.DropoutFn.apply(x, float(p), 10)
