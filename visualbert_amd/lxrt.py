"""The cross-modality building blocks of the sibling model (unsupervised_visualbert/src/lxrt/modeling.py) on the HIP
kernels -- SURVEY 8f / N4, last part.  Same class names, constructor arguments, forward signatures and state-dict keys:

  BertAttention(config, ctx_dim=None)   :349-404   query from hidden_states, key / value from `context` (any width ctx_dim)
  BertAttOutput                         :406-417   dense -> dropout -> LayerNorm(x + input)
  BertCrossattLayer / BertSelfattLayer  :420-443
  BertIntermediate / BertOutput         :445-471
  LXRTXLayer                            :667-717   cross-attention in both directions with ONE shared BertCrossattLayer,
                                                   self-attention per modality, FFN per modality
  VisualFeatEncoder                     :719-767   (LayerNorm(visn_fc(feats)) + LayerNorm(box_fc(boxes))) / 2, dropout

Every FLOP runs in libvisualbert_hip.so: the Linears through vb_gemm (ops.LinearFn), dropout + residual + LayerNorm through
vb_ln_fwd / vb_ln_bwd (ops.LayerNormFn), the attention core through vb_attn_cross_fwd / vb_attn_cross_bwd
(ops.CrossAttentionCoreFn: queries and keys / values from different sequences of different lengths, scores never in HBM).
These are stand-alone modules (no flat parameter arena): gradients reach `.grad` through autograd as usual.

  LXRTEncoder                           :769-905   VisualFeatEncoder + either (visualbert_style) a stack of BertLayer over the
                                                   concatenated [language; vision] sequence, or the l / r / x stack: BertLayer
                                                   over the language, BertLayer over the regions, LXRTXLayer across both

The reference's own constructor reaches the l / r / x stack only through an `assert(0)` (lxrt/modeling.py:803-804): no
argument set builds it there, but its forward (:893-905) is well defined, and tests/golden/base_lxrt_encoder.npz pins this
class against that forward run on the reference's own blocks (the golden generator assembles the module around the assert).  `output_attention` (returning the probabilities) is not offered: NotImplementedError, never silent."""
import torch
from torch import nn

from . import ops
from .modeling import BertConfig, BertLayer, BertLayerNorm  # noqa: F401  (re-exported: the reference module defines them too)


def _p(drop, training):
    return float(drop.p) if training else 0.0


def _mask_add(attention_mask, B, Sk, device):
    """the reference hands an ADDITIVE mask broadcastable to [B, heads, Sq, Sk] (extended attention mask, (1 - m) * -10000);
    the kernels take fp32 [B, Sk] over the keys."""
    if attention_mask is None:
        return torch.zeros((B, Sk), dtype=torch.float32, device=device)
    m = attention_mask.to(torch.float32)
    if m.numel() != B * Sk:
        raise NotImplementedError("attention_mask must broadcast over heads and queries ([B, 1, 1, Sk]); got %s" % (tuple(m.shape),))
    return m.reshape(B, Sk).contiguous()


class BertAttention(nn.Module):
    _next_sid = [64]                      # dropout stream ids of these stand-alone blocks (the encoder layers use 8 .. 8 + 8 L)

    def __init__(self, config, ctx_dim=None):
        super(BertAttention, self).__init__()
        if config.hidden_size % config.num_attention_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (config.hidden_size, config.num_attention_heads))
        self.num_attention_heads = config.num_attention_heads
        self.attention_head_size = int(config.hidden_size / config.num_attention_heads)
        if self.attention_head_size != 64:
            raise NotImplementedError("attention kernels are built for head size 64 (BERT-base 768 / 12)")
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        if ctx_dim is None:
            ctx_dim = config.hidden_size
        self.query = nn.Linear(config.hidden_size, self.all_head_size)
        self.key = nn.Linear(ctx_dim, self.all_head_size)
        self.value = nn.Linear(ctx_dim, self.all_head_size)
        self.dropout = nn.Dropout(config.attention_probs_dropout_prob)
        self._sid = BertAttention._next_sid[0]
        BertAttention._next_sid[0] += 4

    def forward(self, hidden_states, context, attention_mask=None):
        B, Sk = context.size(0), context.size(1)
        q = ops.LinearFn.apply(hidden_states, self.query.weight, self.query.bias, None, False)
        k = ops.LinearFn.apply(context, self.key.weight, self.key.bias, None, False)
        v = ops.LinearFn.apply(context, self.value.weight, self.value.bias, None, False)
        mask_add = _mask_add(attention_mask, B, Sk, hidden_states.device)
        return ops.CrossAttentionCoreFn.apply(q, k, v, mask_add, self.num_attention_heads, _p(self.dropout, self.training),
                                              self._sid)


class BertAttOutput(nn.Module):
    def __init__(self, config):
        super(BertAttOutput, self).__init__()
        self.dense = nn.Linear(config.hidden_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self._sid = BertAttention._next_sid[0]
        BertAttention._next_sid[0] += 4

    def forward(self, hidden_states, input_tensor):
        h = ops.LinearFn.apply(hidden_states, self.dense.weight, self.dense.bias, None, False)
        return ops.LayerNormFn.apply(h, input_tensor, self.LayerNorm.weight, self.LayerNorm.bias,
                                     self.LayerNorm.variance_epsilon, _p(self.dropout, self.training), 0.0, self._sid)


class BertCrossattLayer(nn.Module):
    def __init__(self, config):
        super(BertCrossattLayer, self).__init__()
        self.att = BertAttention(config)
        self.output = BertAttOutput(config)

    def forward(self, input_tensor, ctx_tensor, ctx_att_mask=None):
        output = self.att(input_tensor, ctx_tensor, ctx_att_mask)
        return self.output(output, input_tensor)


class BertSelfattLayer(nn.Module):
    def __init__(self, config):
        super(BertSelfattLayer, self).__init__()
        self.self = BertAttention(config)
        self.output = BertAttOutput(config)

    def forward(self, input_tensor, attention_mask):
        self_output = self.self(input_tensor, input_tensor, attention_mask)      # keys and queries are the same tensor
        return self.output(self_output, input_tensor)


class BertIntermediate(nn.Module):
    def __init__(self, config):
        super(BertIntermediate, self).__init__()
        self.dense = nn.Linear(config.hidden_size, config.intermediate_size)
        if config.hidden_act != "gelu":
            raise NotImplementedError("hidden_act %r: the fused epilogue implements BERT's erf-GELU" % (config.hidden_act,))

    def forward(self, hidden_states):
        return ops.LinearFn.apply(hidden_states, self.dense.weight, self.dense.bias, "gelu", False)


class BertOutput(nn.Module):
    def __init__(self, config):
        super(BertOutput, self).__init__()
        self.dense = nn.Linear(config.intermediate_size, config.hidden_size)
        self.LayerNorm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self._sid = BertAttention._next_sid[0]
        BertAttention._next_sid[0] += 4

    def forward(self, hidden_states, input_tensor):
        h = ops.LinearFn.apply(hidden_states, self.dense.weight, self.dense.bias, None, False)
        return ops.LayerNormFn.apply(h, input_tensor, self.LayerNorm.weight, self.LayerNorm.bias,
                                     self.LayerNorm.variance_epsilon, _p(self.dropout, self.training), 0.0, self._sid)


class LXRTXLayer(nn.Module):
    def __init__(self, config):
        super(LXRTXLayer, self).__init__()
        self.visual_attention = BertCrossattLayer(config)          # ONE set of weights for both directions (:665, :680-681)
        self.lang_self_att = BertSelfattLayer(config)
        self.visn_self_att = BertSelfattLayer(config)
        self.lang_inter = BertIntermediate(config)
        self.lang_output = BertOutput(config)
        self.visn_inter = BertIntermediate(config)
        self.visn_output = BertOutput(config)

    def cross_att(self, lang_input, lang_attention_mask, visn_input, visn_attention_mask):
        lang_att_output = self.visual_attention(lang_input, visn_input, ctx_att_mask=visn_attention_mask)
        visn_att_output = self.visual_attention(visn_input, lang_input, ctx_att_mask=lang_attention_mask)
        return lang_att_output, visn_att_output

    def self_att(self, lang_input, lang_attention_mask, visn_input, visn_attention_mask):
        return self.lang_self_att(lang_input, lang_attention_mask), self.visn_self_att(visn_input, visn_attention_mask)

    def output_fc(self, lang_input, visn_input):
        lang_output = self.lang_output(self.lang_inter(lang_input), lang_input)
        visn_output = self.visn_output(self.visn_inter(visn_input), visn_input)
        return lang_output, visn_output

    def forward(self, lang_feats, lang_attention_mask, visn_feats, visn_attention_mask):
        lang_att_output, visn_att_output = self.cross_att(lang_feats, lang_attention_mask, visn_feats, visn_attention_mask)
        lang_att_output, visn_att_output = self.self_att(lang_att_output, lang_attention_mask, visn_att_output,
                                                         visn_attention_mask)
        return self.output_fc(lang_att_output, visn_att_output)


class VisualFeatEncoder(nn.Module):
    def __init__(self, config, visual_feat_dim=2048, visual_pos_dim=4):
        """the reference reads the two widths from its global VISUAL_CONFIG (:718-719); here they are arguments."""
        super(VisualFeatEncoder, self).__init__()
        self.visn_fc = nn.Linear(visual_feat_dim, config.hidden_size)
        self.visn_layer_norm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.box_fc = nn.Linear(visual_pos_dim, config.hidden_size)
        self.box_layer_norm = BertLayerNorm(config.hidden_size, eps=1e-12)
        self.dropout = nn.Dropout(config.hidden_dropout_prob)
        self._sid = BertAttention._next_sid[0]
        BertAttention._next_sid[0] += 4

    def forward(self, visn_input):
        feats, boxes = visn_input
        dt = feats.dtype
        x = ops.LinearFn.apply(feats, self.visn_fc.weight, self.visn_fc.bias, None, False)
        x = ops.LayerNormFn.apply(x, None, self.visn_layer_norm.weight, self.visn_layer_norm.bias,
                                  self.visn_layer_norm.variance_epsilon, 0.0, 0.0, self._sid)
        # box_fc has K = 4: below the GEMM kernels' 8-element operand granularity -> the box matrix is zero-padded to 8 columns
        # (exact: the padded products are zeros)
        Bq, R, P = boxes.shape
        pad = (-P) % 8
        bx, w = boxes.to(dt), self.box_fc.weight
        if pad:
            bx = torch.nn.functional.pad(bx, (0, pad))
            w = _PadColumns.apply(w, pad)
        y = ops.LinearFn.apply(bx, w, self.box_fc.bias, None, False)
        y = ops.LayerNormFn.apply(y, None, self.box_layer_norm.weight, self.box_layer_norm.bias,
                                  self.box_layer_norm.variance_epsilon, 0.0, 0.0, self._sid + 1)
        out = (x + y) * 0.5
        p = _p(self.dropout, self.training)
        return ops.DropoutFn.apply(out, p, self._sid + 2) if p > 0.0 else out


def _cat_with_none(a, b, dim):
    if a is None:
        return b
    if b is None:
        return a
    return torch.cat((a, b), dim=dim)


class LXRTEncoder(nn.Module):
    """lxrt/modeling.py:769-905.  The reference reads the layer counts and `visualbert_style` from globals (VISUAL_CONFIG /
    args); here they are constructor arguments.  forward(lang_feats, lang_attention_mask, visn_feats, visn_attention_mask):
    lang_feats [B, Tl, H] (the word embeddings are applied outside, :829-830), visn_feats = (features [B, R, Dv], boxes
    [B, R, 4]) or None, masks additive and broadcastable ([B, 1, 1, S], (1 - m) * -10000).  Returns (lang_feats, visn_feats).
    BertLayer is the fused encoder layer of visualbert_amd.modeling (same state-dict keys as the sibling's :719-731)."""

    def __init__(self, config, l_layers=12, x_layers=5, r_layers=0, visualbert_style=False, visual_feat_dim=2048,
                 visual_pos_dim=4):
        super(LXRTEncoder, self).__init__()
        self.visn_fc = VisualFeatEncoder(config, visual_feat_dim=visual_feat_dim, visual_pos_dim=visual_pos_dim)
        self.num_l_layers, self.num_x_layers, self.num_r_layers = l_layers, x_layers, r_layers
        self.visualbert_style = visualbert_style
        self.layer = nn.ModuleList([BertLayer(config) for _ in range(l_layers)])
        for i, m in enumerate(self.layer):                       # dropout stream ids of the fused layers: clear of the blocks'
            m.set_index(64 + i)
        if not visualbert_style:
            self.x_layers = nn.ModuleList([LXRTXLayer(config) for _ in range(x_layers)])
            self.r_layers = nn.ModuleList([BertLayer(config) for _ in range(r_layers)])
            for i, m in enumerate(self.r_layers):
                m.set_index(64 + l_layers + i)
        self.config = config

    def forward(self, lang_feats, lang_attention_mask, visn_feats, visn_attention_mask=None, layer_limit=-1):
        if visn_feats is not None and visn_feats[0] is not None:
            visn_feats = self.visn_fc(visn_feats)
        else:
            visn_feats = None
        if self.visualbert_style:                                # :853-891 (default switches)
            joint = _cat_with_none(lang_feats, visn_feats, dim=1)
            joint_mask = _cat_with_none(lang_attention_mask, visn_attention_mask, dim=-1)
            layers = self.layer if layer_limit == -1 else self.layer[:layer_limit]
            for layer_module in layers:
                joint = layer_module(joint, joint_mask)
            if lang_feats is None:
                return None, joint
            if visn_feats is None:
                return joint, None
            Tl = lang_feats.size(1)
            return joint[:, :Tl, :].contiguous(), joint[:, Tl:, :].contiguous()
        if lang_feats is not None:                               # :893-905
            for layer_module in self.layer:
                lang_feats = layer_module(lang_feats, lang_attention_mask)
        for layer_module in self.r_layers:
            visn_feats = layer_module(visn_feats, visn_attention_mask)
        if lang_feats is not None:
            for layer_module in self.x_layers:
                lang_feats, visn_feats = layer_module(lang_feats, lang_attention_mask, visn_feats, visn_attention_mask)
        return lang_feats, visn_feats


class _PadColumns(torch.autograd.Function):
    """[N, K] -> [N, K + pad] with zero columns, as a fresh fp32 leaf-like tensor the GEMM wrappers accept (they key bf16
    shadows on parameter identity); the gradient of the real columns flows back."""

    @staticmethod
    def forward(ctx, w, pad):
        ctx.k = w.size(1)
        return torch.nn.functional.pad(w.detach(), (0, pad)).contiguous()

    @staticmethod
    def backward(ctx, g):
        return g[:, :ctx.k].contiguous(), None


def lxrt_init_weights(module, initializer_range=0.02):
    """BertPreTrainedModel.init_bert_weights of the sibling (:1105-1116): N(0, 0.02) Linear / Embedding weights, zero biases,
    LayerNorm (1, 0)."""
    if isinstance(module, (nn.Linear, nn.Embedding)):
        module.weight.data.normal_(mean=0.0, std=initializer_range)
    elif isinstance(module, BertLayerNorm):
        module.bias.data.zero_()
        module.weight.data.fill_(1.0)
    if isinstance(module, nn.Linear) and module.bias is not None:
        module.bias.data.zero_()

