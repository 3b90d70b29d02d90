"""Host side of the HIP path: thin wrappers over the C ABI (include/visualbert_hip.h) plus the
torch.autograd.Function objects that stitch the kernels into PyTorch's autograd.

PyTorch is plumbing here (device memory through the caching allocator, streams, autograd graph);
every FLOP of the hot path runs in libvisualbert_hip.so.  There is no eager fallback: a missing
library or a non-zero status raises RuntimeError.

Conventions
  * activations are 2-D [M, features] views, dtype T in {fp32, bf16}; leading dimensions that the
    GEMM reads with 16-byte vectors are multiples of 8 (alloc2d pads).
  * parameters are fp32 masters; in bf16 mode GEMMs read a bf16 shadow copy (`_vb_shadow`) kept
    fresh by the fused optimizer (or re-cast here when the master's version counter moved).
  * parameter gradients: if a parameter carries `_vb_grad` (a view into the flat fp32 gradient arena,
    see modeling.ParameterArena) the kernels ACCUMULATE straight into it and autograd gets None;
    otherwise a fresh fp32 gradient is returned to autograd as usual.
  * dropout masks are regenerated in backward from (seed, stream id); forward draws the seed.
"""
import threading

import torch

from . import _lib
from ._lib import (VB_ACT_GELU, VB_ACT_GELU_GRAD, VB_ACT_GELU_SAVE_GRAD, VB_ACT_MUL_AUX, VB_ACT_NONE, VB_ACT_TANH, VB_KCONTIG, VB_KSTRIDED, check, ptr,
                   stream_ptr)

_ACT = {None: VB_ACT_NONE, "none": VB_ACT_NONE, "gelu": VB_ACT_GELU, "tanh": VB_ACT_TANH}

# ------------------------------------------------------------------------------------------------
# seeds
# ------------------------------------------------------------------------------------------------
_seed_counter = [0]
_seed_replica = [0]


def set_replica(rank):
    """data-parallel rank of this process: replicas draw DIFFERENT dropout masks from the same torch seed (the
    reference's DataParallel replicas share one generator and therefore never repeat each other's masks either)."""
    _seed_replica[0] = int(rank)


def next_seed():
    """64-bit dropout seed: torch's global seed mixed with a call counter and the replica index (deterministic after
    torch.manual_seed for a fixed call order)."""
    _seed_counter[0] += 1
    return (torch.initial_seed() * 0x9E3779B97F4A7C15 + _seed_counter[0] * 0xD1B54A32D192ED03 +
            _seed_replica[0] * 0xA24BAED4963EE407) & 0xFFFFFFFFFFFFFFFF


def reset_seed_counter(v=0):
    _seed_counter[0] = v


# ------------------------------------------------------------------------------------------------
# allocation helpers
# ------------------------------------------------------------------------------------------------
def round_up(x, m):
    return (x + m - 1) // m * m


def alloc2d(M, N, dtype, device, zero=False):
    """[M, N] view whose leading dimension is a multiple of 8 elements."""
    ld = round_up(N, 8)
    buf = (torch.zeros if zero else torch.empty)((M, ld), dtype=dtype, device=device)
    return buf if ld == N else buf[:, :N]


def as2d(x):
    if x.dim() == 2:
        return x
    return x.reshape(-1, x.size(-1))


def _ld(t):
    if t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError("visualbert_amd: expected a 2-D row-major view, got strides %s" % (t.stride(),))
    return t.stride(0)


# ------------------------------------------------------------------------------------------------
# "bf16x3": fp32 GEMMs on the bf16 matrix pipe (include/visualbert_hip.h, VB_BF16X3).  Activations stay fp32 tensors; right
# in front of a GEMM each operand is split into bf16 hi | lo planes and the kernel accumulates hi.hi + lo.hi + hi.lo in fp32.
# The mode is a property of the MODEL (TrainVisualBERTObjective.set_compute_dtype("bf16x3")): its forward runs inside
# x3_scope(True), every autograd Function below remembers the flag for its backward (x3_aware).
# ------------------------------------------------------------------------------------------------
class _X3State(threading.local):
    """per-THREAD mode flag (indexable like the one-element list it replaces): autograd runs backward nodes on its own device
    thread, and a second model may run its forward on another Python thread meanwhile -- neither may see the other's mode.
    Every autograd Function carries the mode on its ctx (x3_aware), so a thread that starts with the default (off) is correct."""

    def __init__(self):
        self.on = False

    def __getitem__(self, i):
        return self.on

    def __setitem__(self, i, v):
        self.on = bool(v)


_x3 = _X3State()
_x3_epoch = [0]


def x3_active():
    return _x3[0]


def bump_x3_epoch():
    """parameters were rewritten behind torch's version counters (fused optimizer step, broadcast, checkpoint restore):
    every cached split weight is stale from here on and is re-split at its next use."""
    _x3_epoch[0] += 1


class x3_scope(object):
    def __init__(self, on):
        self.on = bool(on)

    def __enter__(self):
        self.old = _x3[0]
        _x3[0] = self.on
        return self

    def __exit__(self, *exc):
        _x3[0] = self.old
        return False


def x3_aware(cls):
    """class decorator for the autograd Functions: forward records whether the split-operand mode was on, backward runs under
    the same setting (autograd calls it long after the model's forward scope has closed)."""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *args):
        ctx._vb_x3 = _x3[0]
        return fwd(ctx, *args)

    def backward(ctx, *grads):
        with x3_scope(getattr(ctx, "_vb_x3", False)):
            return bwd(ctx, *grads)

    cls.forward = staticmethod(forward)
    cls.backward = staticmethod(backward)
    return cls


class SplitOperand(object):
    """bf16 [rows, ld] image of an fp32 [rows, cols] matrix: hi plane in columns [0, ld/2), lo plane in [ld/2, ld), pad zeroed.
    Quacks like the fp32 matrix where the callers look at shapes; `master` (optional) is the fp32 tensor it was made from."""

    def __init__(self, buf, cols, master=None):
        self.buf, self.cols, self.master = buf, cols, master
        self.shape = (buf.size(0), cols)
        self.dtype = torch.float32
        self.device = buf.device

    def size(self, i):
        return self.shape[i]

    @property
    def ld(self):
        return self.buf.stride(0)


def split_rows(x, rows, cols, half=None):
    """split `cols` columns of the first `rows` rows of the row-major fp32 matrix behind x (leading dimension x.stride(0); the
    columns may extend into x's zero padding) into a SplitOperand with hi | lo planes of `half` columns (default: cols rounded
    up to 8)."""
    if x.dtype != torch.float32 or x.stride(-1) != 1:
        raise RuntimeError("visualbert_amd.split_rows: fp32 row-major input required")
    half = half or round_up(cols, 8)
    buf = torch.empty((rows, 2 * half), dtype=torch.bfloat16, device=x.device)
    ld_src = x.stride(0) if x.dim() == 2 else cols
    check(_lib.lib().vb_split_bf16(ptr(x), ld_src, ptr(buf), 2 * half, rows, cols, stream_ptr()), "vb_split_bf16")
    return SplitOperand(buf, cols)


def _x3_holder(p):
    return getattr(p, "_o", p)                      # the packed q|k|v alias keeps its cache on the attention module


def _x3_entry(p):
    h = _x3_holder(p)
    e = getattr(h, "_vb_x3_entry", None)
    ver = (p._version, _x3_epoch[0])
    if e is None or e["device"] != p.device:
        e = dict(w=None, wt=None, w_ver=None, wt_ver=None, device=p.device)
        h._vb_x3_entry = e
    return e, ver


def x3_weight(p):
    """split image [N, 2 K'] of the 2-D fp32 weight p [N, K] (cached on the parameter, re-split when it changed)"""
    e, ver = _x3_entry(p)
    w = p.detach()
    N, K = w.shape
    if e["w"] is None:
        e["w"] = SplitOperand(torch.empty((N, 2 * round_up(K, 8)), dtype=torch.bfloat16, device=w.device), K, master=w)
    if e["w_ver"] != ver:
        so = e["w"]
        so.master = w
        check(_lib.lib().vb_split_bf16(ptr(w), w.stride(0), ptr(so.buf), so.ld, N, K, stream_ptr()), "vb_split_bf16")
        e["w_ver"] = ver
    return e["w"]


def x3_weight_t(p):
    """split image of W^T: [K, 2 round_up(N, 64)] for p [N, K] (the dgrad operand; zero padded so the decoder's ragged
    vocabulary reduces over whole K tiles)"""
    e, ver = _x3_entry(p)
    w = p.detach()
    N, K = w.shape
    if e["wt"] is None:
        e["wt"] = SplitOperand(torch.empty((K, 2 * round_up(N, 64)), dtype=torch.bfloat16, device=w.device), N)
    if e["wt_ver"] != ver:
        so = e["wt"]
        check(_lib.lib().vb_split_bf16_t(ptr(w), w.stride(0), ptr(so.buf), so.ld, N, K, stream_ptr()), "vb_split_bf16_t")
        e["wt_ver"] = ver
    return e["wt"]


# ------------------------------------------------------------------------------------------------
# parameters: bf16 shadows and gradient targets
# ------------------------------------------------------------------------------------------------
def weight_for(p, dtype):
    """The tensor a GEMM should read for parameter/buffer `p` when activations have `dtype`."""
    if dtype == torch.float32:
        if _x3[0] and len(p.shape) == 2:
            return x3_weight(p)
        return p.detach()
    sh = getattr(p, "_vb_shadow", None)
    if sh is None or sh.device != p.device:
        sh = torch.empty(p.shape, dtype=torch.bfloat16, device=p.device)
        p._vb_shadow = sh
        p._vb_shadow_ver = -1
    if p._vb_shadow_ver != p._version:
        cast(p.detach(), sh)
        p._vb_shadow_ver = p._version
    return sh


def weight_t_for(p, dtype):
    """W^T (bf16, [in, out] with a padded leading dimension) for dgrad, or None when the parameter has no
    arena-managed transposed shadow (fp32 mode, stand-alone modules): dgrad then reads W K-strided."""
    if dtype == torch.float32 and _x3[0] and len(p.shape) == 2:
        return x3_weight_t(p)
    if dtype != torch.bfloat16:
        return None
    wt = getattr(p, "_vb_shadow_t", None)
    if wt is None:
        return None
    if p._vb_shadow_t_ver != p._version:
        weight_for(p, dtype)                       # bf16 shadow first, then all transposes in one launch
        p._vb_arena.refresh_transposed()
    return wt


def grad_target(p):
    """(fp32 tensor to accumulate into, direct?)"""
    g = getattr(p, "_vb_grad", None)
    if g is not None:
        p._vb_arena.touched.add(id(p))          # "this step wrote a gradient for p" (what `.grad is not None` means upstream)
        return g, True
    return torch.zeros(p.shape, dtype=torch.float32, device=p.device), False


def grad_result(g, direct):
    return None if direct else g


# ------------------------------------------------------------------------------------------------
# raw kernel wrappers
# ------------------------------------------------------------------------------------------------
def cast(src, dst):
    check(_lib.lib().vb_cast(_lib.dtype_code(src.dtype), ptr(src), _lib.dtype_code(dst.dtype), ptr(dst),
                             src.numel(), stream_ptr()), "vb_cast")
    return dst


_profiled_stream = [None]


def gemm_profile_start():
    """HIP-event timing of every GEMM launch enqueued on the CURRENT stream, recorded inside the library (vb_stream_profile)."""
    _profiled_stream[0] = stream_ptr()
    check(_lib.lib().vb_stream_profile(_profiled_stream[0], 1), "vb_stream_profile")


def gemm_profile_stop():
    """call after torch.cuda.synchronize(): {key: dict(ms, flops, launches)}; key = kernel instantiation."""
    import ctypes
    L = _lib.lib()
    cap = 1 << 20
    ms = (ctypes.c_double * cap)()
    fl = (ctypes.c_double * cap)()
    ky = (ctypes.c_int * cap)()
    n = L.vb_stream_profile_read(_profiled_stream[0], ms, fl, ky, cap)
    if n < 0:
        raise RuntimeError("vb_stream_profile_read failed (device not synchronised?)")
    out = {}
    for i in range(n):
        d = out.setdefault(ky[i], dict(ms=0.0, flops=0.0, launches=0))
        d["ms"] += ms[i]
        d["flops"] += fl[i]
        d["launches"] += 1
    L.vb_stream_profile(_profiled_stream[0], 0)
    return out


def gemm_key_name(key):
    if (key & 19) == 19:
        return "gemm_tn_8ph_kernel<bf16->fp32, grouped persistent 256x256 wgrad, ds_read_b64_tr_b16 gathers, fp32 atomics>" + \
            (" [bf16x3: three plane-pair passes, algorithmic FLOPs]" if key & 256 else "")
    if (key & 255) == 39:
        return "gemm_tn_small_kernel<bf16->fp32, grouped 128x128 wgrad for small token counts, whole token range per tile, plain read-modify-write>" + \
            (" [bf16x3: three plane-pair passes, algorithmic FLOPs]" if key & 256 else "")
    if key & 512:
        return "hipBLASLt (vendor yardstick, nt_kernel 200)<bf16->%s>" % ("fp32" if key & 4 else "bf16")
    x3 = " [bf16x3 split operands]" if key & 256 else ""
    if key & 1024:
        x3 += " [K range of each tile split over several workgroups]"
    key &= 255
    if not (key & 3):
        kind = ("gemm_nt_dual_kernel<%s->%s, 256x128 tile, two workgroups per CU, five-slot LDS-direct ring>" if key & 64 else
                "gemm_nt_8ph_kernel<%s->%s, persistent 256x256 tile, four-slot LDS-direct schedule>" if key & 16 else
                "gemm_nt_big_kernel / gemm_nt_bdir_kernel<%s->%s, four waves, 128x128 outputs per wave (developer library)>" if key & 128 else
                "gemm_nt_experimental<%s->%s>" if key & 32 else
                "gemm_nt_pipe_kernel<%s->%s, 256x128 tile, 2-stage LDS-direct>")
        return kind % ("fp32" if key & 8 else "bf16", "fp32" if (key & 4 or key & 8) else "bf16") + x3
    return (x3.strip() + " " if x3 else "") + "gemm_kernel<%s->%s, A %s, B %s>" % ("fp32" if key & 8 else "bf16",
                                                "fp32" if (key & 4 or key & 8) else "bf16",
                                                "Kstrided" if key & 2 else "Kcontig",
                                                "Kstrided" if key & 1 else "Kcontig")


def gemm(a, b, M, N, K, a_layout=VB_KCONTIG, b_layout=VB_KCONTIG, out=None, out_dtype=None, bias=None, act=VB_ACT_NONE,
         addend=None, aux_in=None, aux_out=None, accumulate=False, alpha=1.0, alpha_dev=None, colsum_out=None):
    dt = a.dtype
    if b.dtype != dt:
        raise RuntimeError("visualbert_amd.gemm: operand dtypes differ (%s vs %s)" % (dt, b.dtype))
    if _x3[0] and dt == torch.float32:
        r = _gemm_x3(a, b, M, N, K, a_layout, b_layout, out, out_dtype, bias, act, addend, aux_in, aux_out, accumulate, alpha,
                     alpha_dev, colsum_out)
        if r is not None:
            return r
    if isinstance(a, SplitOperand):                 # a shape the split-operand kernels do not take: the exact fp32 path
        a = a.master
    if isinstance(b, SplitOperand):
        b = b.master
    if a is None or b is None:
        raise RuntimeError("visualbert_amd.gemm: split operand without an fp32 master on a shape that needs the fp32 kernels")
    if out is None:
        out = alloc2d(M, N, out_dtype or dt, a.device)
    aux = aux_in if aux_in is not None else aux_out

    def run():
        return _lib.lib().vb_gemm(_lib.dtype_code(dt), _lib.dtype_code(out.dtype), a_layout, b_layout,
                                  ptr(a), _ld(a), ptr(b), _ld(b), ptr(out), _ld(out), M, N, K, float(alpha),
                                  ptr(alpha_dev), ptr(bias), ptr(addend), _ld(addend) if addend is not None else 0,
                                  act, ptr(aux_in), ptr(aux_out), _ld(aux) if aux is not None else 0,
                                  1 if accumulate else 0, ptr(colsum_out), stream_ptr())
    check(run(), "vb_gemm")
    return out


def _gemm_x3(a, b, M, N, K, a_layout, b_layout, out, out_dtype, bias, act, addend, aux_in, aux_out, accumulate, alpha, alpha_dev,
             colsum_out):
    """the split-operand form of gemm() for fp32 tensors, or None when the shape is not one it takes (ragged K, a K-strided
    operand next to a K-contiguous one): the caller then runs the exact fp32 kernels."""
    L = _lib.lib()
    dev = a.device
    if out is not None and out.dtype != torch.float32:
        return None
    if a_layout == VB_KCONTIG and b_layout == VB_KCONTIG:
        if K % 64:
            return None
        A = a if isinstance(a, SplitOperand) else split_rows(a, M, K)
        Bm = b if isinstance(b, SplitOperand) else split_rows(b, N, K)
        if A.ld // 2 < K or Bm.ld // 2 < K or A.ld % 16 or Bm.ld % 16:
            return None
        if out is None:
            out = alloc2d(M, N, torch.float32, dev)
        aux = aux_in if aux_in is not None else aux_out
        check(L.vb_gemm(_lib.VB_BF16X3, _lib.VB_F32, VB_KCONTIG, VB_KCONTIG, ptr(A.buf), A.ld, ptr(Bm.buf), Bm.ld, ptr(out),
                        _ld(out), M, N, K, float(alpha), ptr(alpha_dev), ptr(bias), ptr(addend),
                        _ld(addend) if addend is not None else 0, act, ptr(aux_in), ptr(aux_out),
                        _ld(aux) if aux is not None else 0, 1 if accumulate else 0, ptr(colsum_out), stream_ptr()), "vb_gemm(bf16x3)")
        return out
    if a_layout == VB_KSTRIDED and b_layout == VB_KSTRIDED and accumulate and out is not None and bias is None and \
            addend is None and act == VB_ACT_NONE and colsum_out is None and not isinstance(a, SplitOperand) and \
            not isinstance(b, SplitOperand):
        # weight gradient dW[M, N] += a^T b over K tokens: a [K, M], b [K, N] token-major
        import ctypes
        A, Bm = split_rows(a, K, M), split_rows(b, K, N)
        P1, I64, I32 = ctypes.c_void_p * 1, ctypes.c_int64 * 1, ctypes.c_int * 1
        check(L.vb_wgrad_grouped(_lib.VB_BF16X3, 1, P1(A.buf.data_ptr()), I64(A.ld), P1(Bm.buf.data_ptr()), I64(Bm.ld),
                                 P1(out.data_ptr()), I64(_ld(out)), I32(M), I32(N), K, float(alpha), ptr(alpha_dev),
                                 stream_ptr()), "vb_wgrad_grouped(bf16x3)")
        return out
    return None


def linear_fwd(x, w, bias, act=VB_ACT_NONE, aux_out=None, out_dtype=None, addend=None):
    """y = act(x w^T + bias); x [M,K] (T), w [N,K] (T)."""
    M, K = x.shape
    N = w.shape[0]
    return gemm(x, w, M, N, K, bias=bias, act=act, aux_out=aux_out, out_dtype=out_dtype, addend=addend)


def linear_dgrad(dy, w, act=VB_ACT_NONE, aux_in=None, addend=None, out=None, alpha_dev=None, wt=None, k_pad=None,
                 colsum_out=None):
    """dx = (dy w) [* gelu'(aux_in)] [+ addend]; dy [M,N], w [N,K].  With wt = w^T ([K, ld >= N]) both operands
    are K-contiguous (LDS-direct loads); otherwise w is read K-strided.  k_pad: reduce over this many
    columns instead of N when BOTH dy and wt are zero-padded that far (ragged vocabulary)."""
    M, N = dy.shape
    K = w.shape[1]
    if isinstance(wt, SplitOperand) and ((k_pad or N) % 64 != 0 or wt.ld // 2 < (k_pad or N)):
        wt = None                                       # not a split-operand shape: exact fp32 kernels on the master weight
    if wt is not None:
        return gemm(dy, wt, M, K, k_pad or N, act=act, aux_in=aux_in, addend=addend, out=out, alpha_dev=alpha_dev,
                    colsum_out=colsum_out)
    return gemm(dy, w, M, K, N, b_layout=VB_KSTRIDED, act=act, aux_in=aux_in, addend=addend, out=out,
                alpha_dev=alpha_dev, colsum_out=colsum_out)


def linear_wgrad(dy, x, dw, alpha_dev=None):
    """dw[N,K] += dy^T x, fp32 accumulate in place; both operands read K-strided."""
    M, N = dy.shape
    K = x.shape[1]
    return gemm(dy, x, N, K, M, a_layout=VB_KSTRIDED, b_layout=VB_KSTRIDED, out=dw, accumulate=True,
                alpha_dev=alpha_dev)


def colsum(x, out, scale_dev=None):
    M, N = x.shape
    check(_lib.lib().vb_colsum(_lib.dtype_code(x.dtype), ptr(x), _ld(x), ptr(out), ptr(scale_dev), M, N,
                               stream_ptr()), "vb_colsum")
    return out


def act_bwd(dy, aux, act):
    dx = torch.empty_like(dy)
    if not (dy.is_contiguous() and aux.is_contiguous()):
        raise RuntimeError("visualbert_amd.act_bwd: contiguous tensors required")
    check(_lib.lib().vb_act_bwd(_lib.dtype_code(dy.dtype), ptr(dy), ptr(aux), ptr(dx), dy.numel(), act, stream_ptr()),
          "vb_act_bwd")
    return dx


def ln_fwd(x, resid, gamma, beta, eps, p_in=0.0, sid_in=0, p_out=0.0, sid_out=0, seed=0, save_z=True, save_stats=True):
    M, H = x.shape
    y = torch.empty((M, H), dtype=x.dtype, device=x.device)
    z = torch.empty((M, H), dtype=x.dtype, device=x.device) if save_z else None
    mean = torch.empty(M, dtype=torch.float32, device=x.device) if save_stats else None
    rstd = torch.empty(M, dtype=torch.float32, device=x.device) if save_stats else None
    if not x.is_contiguous() or (resid is not None and not resid.is_contiguous()):
        raise RuntimeError("visualbert_amd.ln_fwd: contiguous [M,H] inputs required")
    check(_lib.lib().vb_ln_fwd(_lib.dtype_code(x.dtype), ptr(x), ptr(resid), ptr(z), ptr(y), ptr(mean), ptr(rstd),
                               ptr(gamma), ptr(beta), M, H, float(eps), float(p_in), sid_in, float(p_out), sid_out,
                               seed, stream_ptr()), "vb_ln_fwd")
    return y, z, mean, rstd


def ln_bwd(dy, z, mean, rstd, gamma, dgamma, dbeta, dbias=None, p_in=0.0, sid_in=0, p_out=0.0, sid_out=0, seed=0):
    """returns (dz, dx); dx is dz itself when p_in == 0."""
    M, H = dy.shape
    if not dy.is_contiguous():
        dy = dy.contiguous()
    dz = torch.empty((M, H), dtype=dy.dtype, device=dy.device)
    dx = torch.empty((M, H), dtype=dy.dtype, device=dy.device) if p_in > 0.0 else dz
    ws = torch.empty(_lib.lib().vb_ln_bwd_ws_bytes(M, H) // 4, dtype=torch.float32, device=dy.device)
    check(_lib.lib().vb_ln_bwd(_lib.dtype_code(dy.dtype), ptr(dy), ptr(z), ptr(mean), ptr(rstd), ptr(gamma), ptr(dz),
                               ptr(dx), ptr(dgamma), ptr(dbeta), ptr(dbias), M, H, float(p_in), sid_in, float(p_out),
                               sid_out, seed, ptr(ws), stream_ptr()), "vb_ln_bwd")
    return dz, dx


def _attn_code(dt):
    """dtype code of the attention kernels: fp32 tensors in split mode run the three-product bf16 MFMA form."""
    return _lib.VB_BF16X3 if (_x3[0] and dt == torch.float32) else _lib.dtype_code(dt)


def attn_fwd(qkv, mask_add, B, S, nh, p, seed, sid):
    H = nh * 64
    ctx = torch.empty((B * S, H), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B, nh, S), dtype=torch.float32, device=qkv.device)
    bits = None
    if p > 0.0:
        nwords = _lib.lib().vb_attn_keepbits_words(S)
        bits = torch.empty(B * nh * nwords, dtype=torch.int64, device=qkv.device)
    check(_lib.lib().vb_attn_fwd(_attn_code(qkv.dtype), ptr(qkv), ptr(mask_add), ptr(ctx), ptr(lse), ptr(bits),
                                 B, S, nh, 64, float(p), seed, sid, stream_ptr()), "vb_attn_fwd")
    return ctx, lse, bits


def attn_bwd(qkv, mask_add, dctx, lse, bits, B, S, nh, p, seed, sid, ctx_fwd=None, dqkv_bias=None):
    """ctx_fwd: the forward output (enables the one-pass bf16 backward, see vb_attn_bwd); dqkv_bias: fp32 [3H] that
    receives += the column sums of dqkv (the packed q | k | v bias gradient)."""
    dqkv = torch.empty_like(qkv)
    ws = torch.empty(_lib.lib().vb_attn_bwd_ws_floats(B, S, nh), dtype=torch.float32, device=qkv.device)
    if not dctx.is_contiguous():
        dctx = dctx.contiguous()
    check(_lib.lib().vb_attn_bwd(_attn_code(qkv.dtype), ptr(qkv), ptr(mask_add), ptr(dctx), ptr(lse), ptr(bits),
                                 ptr(ws), ptr(dqkv), ptr(ctx_fwd), ptr(dqkv_bias), B, S, nh, 64, float(p), seed, sid,
                                 stream_ptr()),
          "vb_attn_bwd")
    return dqkv


def attention_probs(h, attn_self, mask_add):
    """softmax(QK^T / sqrt(d) + mask) as fp32 [B, nh, S, S] -- what BertSelfAttention returns next to the context
    under output_attention_weights (modeling.py:241-261).  Forward only: the packed QKV projection is recomputed
    here because the training path never materialises the probabilities."""
    B, S, H = h.shape
    h2 = h.detach().reshape(B * S, H)
    if not h2.is_contiguous():
        h2 = h2.contiguous()
    wqkv, bqkv = _packed_qkv(attn_self, h2.dtype)
    qkv = linear_fwd(h2, wqkv, bqkv)
    nh = attn_self.num_attention_heads
    probs = torch.empty((B, nh, S, S), dtype=torch.float32, device=h.device)
    check(_lib.lib().vb_attn_probs(_lib.dtype_code(qkv.dtype), ptr(qkv), ptr(mask_add), ptr(probs), B, S, nh, H // nh,
                                   stream_ptr()), "vb_attn_probs")
    return probs


def prepare_inputs(input_mask, image_dim, image_mask, lm_labels, R):
    B, T = input_mask.shape
    dev = input_mask.device
    S = T + R
    am = torch.empty((B, S), dtype=torch.int64, device=dev)
    ma = torch.empty((B, S), dtype=torch.float32, device=dev)
    le = torch.empty((B, S), dtype=torch.int64, device=dev) if lm_labels is not None else None
    # contiguous copies are bound to names: a temporary would be released (and its block possibly reused by the next
    # allocation) before the launch that reads it is enqueued
    im_c = input_mask.contiguous()
    lab_c = lm_labels.contiguous() if lm_labels is not None else None
    check(_lib.lib().vb_prepare_inputs(ptr(im_c), ptr(image_dim), ptr(image_mask),
                                       ptr(lab_c), ptr(am), ptr(ma),
                                       ptr(le), B, T, R, stream_ptr()), "vb_prepare_inputs")
    return am, ma, le


def _upstream_scalar(g):
    """fp32 device scalar carrying d(total)/d(loss); None when it is exactly the implicit 1."""
    if g is None:
        return None
    return g.detach().reshape(1).to(torch.float32).contiguous()


# ------------------------------------------------------------------------------------------------
# masked-row plan of the MLM head: which rows carry a label, known to the HOST without draining the GPU queue.
# The count is launched where the labels first appear (start of the model forward) and copied to pinned memory
# asynchronously; by the time the head needs it (12 layers later) the copy has long finished -- and if the host is
# a step ahead, waiting on its event leaves the whole forward queued behind it, so the GPU never idles.
# ------------------------------------------------------------------------------------------------
class MaskedRowPlan:
    def __init__(self, labels):
        lab = labels.reshape(-1)
        self.key = (lab.data_ptr(), lab.numel())
        self.valid = lab != -1
        self.count_host = None
        self.event = None
        if lab.is_cuda:
            cnt = self.valid.sum(dtype=torch.int64)
            self.count_host = torch.empty(1, dtype=torch.int64, pin_memory=True)
            self.count_host.copy_(cnt, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()

    def rows(self):
        """int64 indices of the labelled rows (ascending)."""
        if self.event is None:
            return torch.nonzero(self.valid).reshape(-1)
        self.event.synchronize()
        n = int(self.count_host[0])
        if n == 0:
            return torch.empty(0, dtype=torch.int64, device=self.valid.device)
        try:
            return torch.nonzero_static(self.valid, size=n).reshape(-1)       # sized on the host: no sync
        except (RuntimeError, NotImplementedError):
            return torch.nonzero(self.valid).reshape(-1)


_row_plan = None


def plan_masked_rows(labels):
    """call as early as the (extended) MLM labels exist; MLMHeadLossFn picks the plan up by tensor identity."""
    global _row_plan
    _row_plan = MaskedRowPlan(labels) if labels is not None else None


def _rows_for(lab):
    global _row_plan
    plan, _row_plan = _row_plan, None
    if plan is not None and plan.key == (lab.data_ptr(), lab.numel()):
        return plan.rows()
    return torch.nonzero(lab != -1).reshape(-1)           # no plan: one host sync


# ------------------------------------------------------------------------------------------------
# autograd Functions
# ------------------------------------------------------------------------------------------------
@x3_aware
class LinearFn(torch.autograd.Function):
    """nn.Linear (+ GELU / tanh) -- modeling.py:232-234, 271, 303-304, 316, 383-385, 398-399, 1220."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, out_fp32):
        x2 = as2d(x)
        if not x2.is_contiguous() and x2.stride(1) != 1:
            x2 = x2.contiguous()
        w = weight_for(weight, x2.dtype)
        code = _ACT[act]
        pre = None
        if code == VB_ACT_GELU:
            pre = alloc2d(x2.size(0), w.size(0), x2.dtype, x2.device)
        y = linear_fwd(x2, w, bias.detach() if bias is not None else None, code, aux_out=pre,
                       out_dtype=torch.float32 if out_fp32 else None)
        ctx.act = code
        ctx.weight, ctx.bias = weight, bias
        ctx.x_shape = x.shape
        ctx.save_for_backward(x2, pre if code == VB_ACT_GELU else (y if code == VB_ACT_TANH else None))
        return y.reshape(*x.shape[:-1], w.size(0)) if y.is_contiguous() else y.view(*x.shape[:-1], w.size(0))

    @staticmethod
    def backward(ctx, dy):
        x2, aux = ctx.saved_tensors
        dy2 = as2d(dy)
        if dy2.dtype != x2.dtype:
            dy2 = dy2.to(x2.dtype)
        if dy2.stride(1) != 1 or (dy2.stride(0) % 8) != 0 or (dy2.data_ptr() % 16) != 0:
            t = alloc2d(dy2.size(0), dy2.size(1), dy2.dtype, dy2.device, zero=True)
            t.copy_(dy2)
            dy2 = t
        if ctx.act in (VB_ACT_GELU, VB_ACT_TANH):
            if not (dy2.is_contiguous() and aux.is_contiguous()):
                dyc, auxc = dy2.contiguous(), aux.contiguous()
                d = act_bwd(dyc, auxc, ctx.act)
                t = alloc2d(d.size(0), d.size(1), d.dtype, d.device, zero=True)
                t.copy_(d)
                dy2 = t
            else:
                dy2 = act_bwd(dy2, aux, ctx.act)
        w = weight_for(ctx.weight, x2.dtype)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = linear_dgrad(dy2, w, wt=weight_t_for(ctx.weight, x2.dtype)).reshape(ctx.x_shape)
        gw, direct_w = grad_target(ctx.weight)
        linear_wgrad(dy2, x2, gw)
        gb_out = None
        if ctx.bias is not None:
            gb, direct_b = grad_target(ctx.bias)
            colsum(dy2, gb)
            gb_out = grad_result(gb, direct_b)
        return dx, grad_result(gw, direct_w), gb_out, None, None


def dropout_apply(x, p, seed, sid):
    """y = dropout(x) with the counter-based mask of (seed, sid); the same call on a gradient is the backward."""
    xc = x.contiguous()
    y = torch.empty_like(xc)
    check(_lib.lib().vb_dropout(_lib.dtype_code(xc.dtype), ptr(xc), ptr(y), xc.numel(), float(p), seed, sid, stream_ptr()),
          "vb_dropout")
    return y


@x3_aware
class DropoutFn(torch.autograd.Function):
    """nn.Dropout in front of the fine-tuning heads (modeling.py:1495, 1557) without a mask tensor."""

    @staticmethod
    def forward(ctx, x, p, sid):
        seed = next_seed()
        ctx.cfg = (p, seed, sid)
        return dropout_apply(x, p, seed, sid).view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        p, seed, sid = ctx.cfg
        return dropout_apply(dy, p, seed, sid).view(dy.shape), None, None


@x3_aware
class LayerNormFn(torch.autograd.Function):
    """y = dropout_out(LN(dropout_in(x) + resid)) -- modeling.py:171-175 with :272-273 / :317-318 / :1255-1256."""

    @staticmethod
    def forward(ctx, x, resid, gamma, beta, eps, p_in, p_out, sid):
        x2 = as2d(x).contiguous()
        r2 = as2d(resid).contiguous() if resid is not None else None
        seed = next_seed()
        need_z = (r2 is not None) or p_in > 0.0
        y, z, mean, rstd = ln_fwd(x2, r2, gamma.detach(), beta.detach(), eps, p_in, sid, p_out, sid + 1, seed,
                                  save_z=need_z)
        ctx.cfg = (p_in, p_out, sid, seed, resid is not None)
        ctx.gamma, ctx.beta = gamma, beta
        ctx.save_for_backward(z if need_z else x2, mean, rstd)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        z, mean, rstd = ctx.saved_tensors
        p_in, p_out, sid, seed, has_resid = ctx.cfg
        dy2 = as2d(dy)
        if dy2.dtype != z.dtype:
            dy2 = dy2.to(z.dtype)
        gg, dg = grad_target(ctx.gamma)
        gb, db = grad_target(ctx.beta)
        dz, dx = ln_bwd(dy2, z, mean, rstd, ctx.gamma.detach(), gg, gb, None, p_in, sid, p_out, sid + 1, seed)
        return (dx.view(dy.shape), dz.view(dy.shape) if has_resid else None, grad_result(gg, dg), grad_result(gb, db),
                None, None, None, None)


@x3_aware
class SelfAttentionCoreFn(torch.autograd.Function):
    """packed qkv [B,S,3H] -> context [B,S,H]; modeling.py:236-256."""

    @staticmethod
    def forward(ctx, qkv, mask_add, nh, p, sid):
        B, S, H3 = qkv.shape
        q2 = qkv.reshape(B * S, H3)
        if not q2.is_contiguous():
            q2 = q2.contiguous()
        seed = next_seed()
        c, lse, bits = attn_fwd(q2, mask_add, B, S, nh, p, seed, sid)
        ctx.cfg = (B, S, nh, p, seed, sid)
        ctx.bits = bits
        ctx.save_for_backward(q2, mask_add, lse, c)
        return c.view(B, S, H3 // 3)

    @staticmethod
    def backward(ctx, dctx):
        q2, mask_add, lse, c = ctx.saved_tensors
        B, S, nh, p, seed, sid = ctx.cfg
        d2 = dctx.reshape(B * S, -1)
        if d2.dtype != q2.dtype:
            d2 = d2.to(q2.dtype)
        dqkv = attn_bwd(q2, mask_add, d2, lse, ctx.bits, B, S, nh, p, seed, sid, ctx_fwd=c)
        return dqkv.view(B, S, -1), None, None, None, None


@x3_aware
class CrossAttentionCoreFn(torch.autograd.Function):
    """softmax(Q K^T / 8 + mask) V with queries and keys / values from DIFFERENT sequences: q [B, Sq, H], k, v [B, Sk, H]
    (any row pitch), mask_add fp32 [B, Sk] over the keys -> context [B, Sq, H].  The core of the LXRT sibling's
    BertAttention(hidden_states, context) (unsupervised_visualbert/src/lxrt/modeling.py:377-411)."""

    @staticmethod
    def forward(ctx, q, k, v, mask_add, nh, p, sid):
        B, Sq, H = q.shape
        Sk = k.size(1)
        q2, k2, v2 = (t.reshape(-1, H) if t.is_contiguous() else t.contiguous().reshape(-1, H) for t in (q, k, v))
        dt = q2.dtype
        out = torch.empty((B * Sq, H), dtype=dt, device=q2.device)
        lse = torch.empty((B, nh, Sq), dtype=torch.float32, device=q2.device)
        L = _lib.lib()
        bits = None
        if p > 0.0:
            bits = torch.empty(B * nh * L.vb_attn_cross_keepbits_words(Sq, Sk), dtype=torch.int64, device=q2.device)
        seed = next_seed()
        check(L.vb_attn_cross_fwd(_attn_code(dt), ptr(q2), _ld(q2), ptr(k2), _ld(k2), ptr(v2), _ld(v2), ptr(mask_add),
                                  ptr(out), _ld(out), ptr(lse), ptr(bits), B, Sq, Sk, nh, 64, float(p), seed, sid,
                                  stream_ptr()), "vb_attn_cross_fwd")
        ctx.cfg = (B, Sq, Sk, H, nh, p, seed, sid)
        ctx.bits = bits
        ctx.save_for_backward(q2, k2, v2, mask_add, lse)
        return out.view(B, Sq, H)

    @staticmethod
    def backward(ctx, dctx):
        q2, k2, v2, mask_add, lse = ctx.saved_tensors
        B, Sq, Sk, H, nh, p, seed, sid = ctx.cfg
        d2 = dctx.reshape(B * Sq, H)
        if d2.dtype != q2.dtype:
            d2 = d2.to(q2.dtype)
        if not d2.is_contiguous():
            d2 = d2.contiguous()
        dq, dk, dv = torch.empty_like(q2), torch.empty_like(k2), torch.empty_like(v2)
        ws = torch.empty((B, nh, Sq), dtype=torch.float32, device=q2.device)
        check(_lib.lib().vb_attn_cross_bwd(_attn_code(q2.dtype), ptr(q2), _ld(q2), ptr(k2), _ld(k2), ptr(v2), _ld(v2),
                                           ptr(mask_add), ptr(d2), _ld(d2), ptr(lse), ptr(ctx.bits), ptr(ws), ptr(dq), _ld(dq),
                                           ptr(dk), _ld(dk), ptr(dv), _ld(dv), B, Sq, Sk, nh, 64, float(p), seed, sid,
                                           stream_ptr()), "vb_attn_cross_bwd")
        return dq.view(B, Sq, H), dk.view(B, Sk, H), dv.view(B, Sk, H), None, None, None, None


def _packed_qkv(attn_self, dtype):
    """(weight [3H,H] in `dtype`, bias [3H] fp32) of a BertSelfAttention with packed storage."""
    w = weight_for(attn_self.qkv_weight, dtype)
    return w, attn_self.qkv_bias.detach()


@x3_aware
class AttentionBlockFn(torch.autograd.Function):
    """BertAttention = BertSelfAttention + BertSelfOutput, fused: packed QKV GEMM -> attention ->
    output GEMM -> dropout + residual + LayerNorm (modeling.py:231-274).  Backward runs
    LN' -> wgrad/dgrad -> attention' -> wgrad/dgrad(+ residual gradient as GEMM addend)."""

    @staticmethod
    def forward(ctx, h, mask_add, module, p_hidden, p_attn, sid, *params):
        B, S, H = h.shape
        sa, so = module.self, module.output
        h2 = h.reshape(B * S, H)
        if not h2.is_contiguous():
            h2 = h2.contiguous()
        dt = h2.dtype
        wqkv, bqkv = _packed_qkv(sa, dt)
        qkv = linear_fwd(h2, wqkv, bqkv)
        seed = next_seed()
        c, lse, bits = attn_fwd(qkv, mask_add, B, S, sa.num_attention_heads, p_attn, seed, sid)
        ao = linear_fwd(c, weight_for(so.dense.weight, dt), so.dense.bias.detach())
        y, z, mean, rstd = ln_fwd(ao, h2, so.LayerNorm.weight.detach(), so.LayerNorm.bias.detach(),
                                  so.LayerNorm.variance_epsilon, p_hidden, sid + 1, 0.0, 0, seed)
        ctx.module = module
        ctx.cfg = (B, S, H, p_hidden, p_attn, sid, seed)
        ctx.bits = bits
        ctx.save_for_backward(h2, mask_add, qkv, c, lse, z, mean, rstd)
        return y.view(B, S, H)

    @staticmethod
    def backward(ctx, dy):
        h2, mask_add, qkv, c, lse, z, mean, rstd = ctx.saved_tensors
        B, S, H, p_hidden, p_attn, sid, seed = ctx.cfg
        sa, so = ctx.module.self, ctx.module.output
        dt = h2.dtype
        dy2 = dy.reshape(B * S, H)
        if dy2.dtype != dt:
            dy2 = dy2.to(dt)
        g_ln_w, d1 = grad_target(so.LayerNorm.weight)
        g_ln_b, d2 = grad_target(so.LayerNorm.bias)
        g_ob, d3 = grad_target(so.dense.bias)
        dz, dao = ln_bwd(dy2, z, mean, rstd, so.LayerNorm.weight.detach(), g_ln_w, g_ln_b, g_ob, p_hidden, sid + 1,
                         0.0, 0, seed)
        g_ow, d4 = grad_target(so.dense.weight)
        linear_wgrad(dao, c, g_ow)
        dctx = linear_dgrad(dao, weight_for(so.dense.weight, dt))
        g_qkv_w, g_qkv_b, direct_qkv = sa.qkv_grad_targets()
        dqkv = attn_bwd(qkv, mask_add, dctx, lse, ctx.bits, B, S, sa.num_attention_heads, p_attn, seed, sid, ctx_fwd=c,
                        dqkv_bias=g_qkv_b)
        linear_wgrad(dqkv, h2, g_qkv_w)
        wqkv, _ = _packed_qkv(sa, dt)
        dh = linear_dgrad(dqkv, wqkv, addend=dz)          # + gradient of the residual connection
        if direct_qkv:
            gq = [None] * 6
        else:
            gw, gb = g_qkv_w.view(3, H, H), g_qkv_b.view(3, H)
            gq = [gw[0], gb[0], gw[1], gb[1], gw[2], gb[2]]
        return (dh.view(B, S, H), None, None, None, None, None, *gq,
                grad_result(g_ow, d4), grad_result(g_ob, d3), grad_result(g_ln_w, d1), grad_result(g_ln_b, d2))


@x3_aware
class FFNBlockFn(torch.autograd.Function):
    """BertIntermediate + BertOutput fused (modeling.py:302-305, 315-319): GEMM+bias+erf-GELU epilogue,
    GEMM+bias, dropout + residual + LayerNorm; backward folds GELU' into the dgrad GEMM epilogue."""

    @staticmethod
    def forward(ctx, a, inter_mod, out_mod, p_hidden, sid, *params):
        B, S, H = a.shape
        a2 = a.reshape(B * S, H)
        if not a2.is_contiguous():
            a2 = a2.contiguous()
        dt = a2.dtype
        I = inter_mod.dense.weight.size(0)
        pre = torch.empty((B * S, I), dtype=dt, device=a2.device)
        inter = linear_fwd(a2, weight_for(inter_mod.dense.weight, dt), inter_mod.dense.bias.detach(),
                           VB_ACT_GELU_SAVE_GRAD, aux_out=pre)      # pre <- gelu'(pre-activation)
        fo = linear_fwd(inter, weight_for(out_mod.dense.weight, dt), out_mod.dense.bias.detach())
        seed = next_seed()
        y, z, mean, rstd = ln_fwd(fo, a2, out_mod.LayerNorm.weight.detach(), out_mod.LayerNorm.bias.detach(),
                                  out_mod.LayerNorm.variance_epsilon, p_hidden, sid, 0.0, 0, seed)
        ctx.mods = (inter_mod, out_mod)
        ctx.cfg = (B, S, H, p_hidden, sid, seed)
        ctx.save_for_backward(a2, pre, inter, z, mean, rstd)
        return y.view(B, S, H)

    @staticmethod
    def backward(ctx, dy):
        a2, pre, inter, z, mean, rstd = ctx.saved_tensors
        B, S, H, p_hidden, sid, seed = ctx.cfg
        im, om = ctx.mods
        dt = a2.dtype
        dy2 = dy.reshape(B * S, H)
        if dy2.dtype != dt:
            dy2 = dy2.to(dt)
        g_ln_w, d1 = grad_target(om.LayerNorm.weight)
        g_ln_b, d2 = grad_target(om.LayerNorm.bias)
        g_ob, d3 = grad_target(om.dense.bias)
        dz, dfo = ln_bwd(dy2, z, mean, rstd, om.LayerNorm.weight.detach(), g_ln_w, g_ln_b, g_ob, p_hidden, sid, 0.0, 0,
                         seed)
        g_ow, d4 = grad_target(om.dense.weight)
        linear_wgrad(dfo, inter, g_ow)
        g_ib, d5 = grad_target(im.dense.bias)
        dpre = linear_dgrad(dfo, weight_for(om.dense.weight, dt), act=VB_ACT_MUL_AUX, aux_in=pre,
                            wt=weight_t_for(om.dense.weight, dt), colsum_out=g_ib)
        g_iw, d6 = grad_target(im.dense.weight)
        linear_wgrad(dpre, a2, g_iw)
        da = linear_dgrad(dpre, weight_for(im.dense.weight, dt), addend=dz)
        return (da.view(B, S, H), None, None, None, None,
                grad_result(g_iw, d6), grad_result(g_ib, d5), grad_result(g_ow, d4), grad_result(g_ob, d3),
                grad_result(g_ln_w, d1), grad_result(g_ln_b, d2))


_scratch = {}


def layer_scratch(nbytes, device):
    """one reusable scratch buffer per device (all layers run on one stream, strictly in sequence)."""
    buf = _scratch.get(device)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _scratch[device] = buf
    return buf


def _ptr_array(items):
    import ctypes
    arr = (ctypes.c_void_p * len(items))()
    for i, t in enumerate(items):
        arr[i] = (t.buf if isinstance(t, SplitOperand) else t).data_ptr()
    return arr


# ---- per-layer call plans (round 6) ---------------------------------------------------------------------------------------
# At per-GPU batches of 8-16 the kernels of a step take 4.7 ms and the HOST needs 5.5 ms to enqueue them (profiles/r06_host_profile_b8.txt):
# most of a BertLayer call was Python plumbing -- nn.Module attribute chains (1 500 __getattr__ per step), shadow / version checks per
# weight, ctypes arrays rebuilt from scratch.  A layer whose 16 parameters live in a ParameterArena (bf16 mode) gets its pointer arrays
# built ONCE and reused while a cheap signature holds: the identity of the sub-modules and parameters (their _modules / _parameters
# dictionaries are consulted directly), every parameter's torch version counter (a write through torch moves it: the slow path then
# refreshes the shadows and rebuilds the plan), and the arena's data / gradient addresses (an arena rebuilt by .to(device) moves them).
class _LayerPlan(object):
    __slots__ = ("sig", "w_arr", "g_arr", "wt_arr", "ld_arr", "touched", "arena", "keep")


def layer_params(layer):
    """((attention, self-attention, self-output, intermediate, output), the 8 parameter-owning modules, their 16 parameters in the
    order BertLayerFn takes them) -- cached on the layer, revalidated by identity against the modules' own dictionaries"""
    c = layer.__dict__.get("_vb_pcache")
    if c is not None:
        (at, sa, so, im, om), mods, params = c
        lm = layer._modules
        ok = lm["attention"] is at and lm["intermediate"] is im and lm["output"] is om and at._modules["self"] is sa and \
            at._modules["output"] is so and sa._modules["query"] is mods[0] and sa._modules["key"] is mods[1] and \
            sa._modules["value"] is mods[2] and so._modules["dense"] is mods[3] and so._modules["LayerNorm"] is mods[4] and \
            im._modules["dense"] is mods[5] and om._modules["dense"] is mods[6] and om._modules["LayerNorm"] is mods[7]
        if ok:
            i = 0
            for m in mods:
                pd = m._parameters
                if pd["weight"] is not params[i] or pd["bias"] is not params[i + 1]:
                    ok = False
                    break
                i += 2
        if ok:
            return c
    at, im, om = layer.attention, layer.intermediate, layer.output
    sa, so = at.self, at.output
    mods = (sa.query, sa.key, sa.value, so.dense, so.LayerNorm, im.dense, om.dense, om.LayerNorm)
    params = tuple(x for m in mods for x in (m.weight, m.bias))
    c = ((at, sa, so, im, om), mods, params)
    layer.__dict__["_vb_pcache"] = c
    return c


def _layer_plan(layer, pc, dt):
    """the cached pointer arrays of an arena-managed bf16 layer, or None (any other configuration takes the general path)"""
    if dt != torch.bfloat16 or _x3[0]:
        return None
    params = pc[2]
    qw = params[0]
    g0 = getattr(qw, "_vb_grad", None)
    if g0 is None:
        return None
    sig = (qw.data_ptr(), g0.data_ptr()) + tuple([p._version for p in params])
    plan = layer.__dict__.get("_vb_plan")
    if plan is not None and plan.sig == sig:
        return plan
    import ctypes
    (at, sa, so, im, om), mods, _ = pc
    arena = getattr(qw, "_vb_arena", None)
    if arena is None or any(getattr(p, "_vb_grad", None) is None or getattr(p, "_vb_arena", None) is not arena for p in params):
        return None
    # the general path's helpers refresh stale shadows as a side effect -- run them once, then take the addresses
    weights = [weight_for(sa.qkv_weight, dt), sa.qkv_bias, weight_for(so.dense.weight, dt), so.dense.bias.detach(),
               so.LayerNorm.weight.detach(), so.LayerNorm.bias.detach(),
               weight_for(im.dense.weight, dt), im.dense.bias.detach(),
               weight_for(om.dense.weight, dt), om.dense.bias.detach(),
               om.LayerNorm.weight.detach(), om.LayerNorm.bias.detach()]
    wts = [weight_t_for(sa.qkv_weight, dt), weight_t_for(so.dense.weight, dt), weight_t_for(im.dense.weight, dt),
           weight_t_for(om.dense.weight, dt)]
    g_qkv_w, g_qkv_b, direct_qkv = sa.qkv_grad_targets()
    if not direct_qkv or any(w is None or isinstance(w, SplitOperand) for w in wts):
        return None
    rest = params[6:]
    grads = [g_qkv_w, g_qkv_b] + [p._vb_grad for p in rest]
    plan = _LayerPlan()
    plan.w_arr = _ptr_array(weights)
    plan.g_arr = _ptr_array(grads)
    plan.wt_arr = (ctypes.c_void_p * 4)(*[w.data_ptr() for w in wts])
    plan.ld_arr = (ctypes.c_int64 * 4)(*[w.stride(0) for w in wts])
    plan.touched = frozenset(id(p) for p in params)
    plan.arena = arena
    plan.keep = (weights, wts, grads)                       # the views the raw pointers came from
    plan.sig = (qw.data_ptr(), g0.data_ptr()) + tuple([p._version for p in params])   # (the helpers may have moved nothing; re-read anyway)
    layer.__dict__["_vb_plan"] = plan
    return plan


_layer_sizes = {}
_NONE16 = (None,) * 16


def _layer_bytes(L, code, B, S, H, I, nh, p_attn):
    key = (code, B, S, H, I, nh, p_attn)
    v = _layer_sizes.get(key)
    if v is None:
        v = (L.vb_bert_layer_saved_bytes(code, B, S, H, I, nh, float(p_attn)), L.vb_bert_layer_scratch_bytes(code, B, S, H, I, nh))
        _layer_sizes[key] = v
    return v


@x3_aware
class BertLayerFn(torch.autograd.Function):
    """A whole BertLayer (modeling.py:331-341) as ONE autograd node: forward and backward are one C-ABI
    call each (vb_bert_layer_fwd / vb_bert_layer_bwd, csrc/layer.hip sequences the 7 + 15 launches)."""

    @staticmethod
    def forward(ctx, h, mask_add, layer, p_hidden, p_attn, *params):
        B, S, H = h.shape
        pc = layer_params(layer)
        (at, sa, so, im, om), mods, _ = pc
        h2 = h.reshape(B * S, H)
        if not h2.is_contiguous():
            h2 = h2.contiguous()
        dt = h2.dtype
        code = _lib.VB_BF16X3 if (_x3[0] and dt == torch.float32) else _lib.dtype_code(dt)
        I = pc[2][10].size(0)                                  # intermediate.dense.weight
        nh = sa.num_attention_heads
        L = _lib.lib()
        nsaved, nscr = _layer_bytes(L, code, B, S, H, I, nh, p_attn)
        if nsaved < 0 or nscr < 0:
            raise RuntimeError("visualbert_amd: unsupported BertLayer shape B=%d S=%d H=%d I=%d heads=%d" % (B, S, H, I, nh))
        saved = torch.empty(nsaved, dtype=torch.uint8, device=h2.device)
        scratch = layer_scratch(nscr, h2.device)
        out = torch.empty((B * S, H), dtype=dt, device=h2.device)
        plan = _layer_plan(layer, pc, dt)
        if plan is not None:
            w_arr = plan.w_arr
        else:
            wqkv = weight_for(sa.qkv_weight, dt)
            weights = [wqkv, sa.qkv_bias, weight_for(so.dense.weight, dt), so.dense.bias.detach(),
                       so.LayerNorm.weight.detach(), so.LayerNorm.bias.detach(),
                       weight_for(im.dense.weight, dt), im.dense.bias.detach(),
                       weight_for(om.dense.weight, dt), om.dense.bias.detach(),
                       om.LayerNorm.weight.detach(), om.LayerNorm.bias.detach()]
            w_arr = _ptr_array(weights)
        seed = next_seed()
        sid = at._sid
        check(L.vb_bert_layer_fwd(code, ptr(h2), ptr(mask_add), ptr(out), ptr(saved), ptr(scratch), w_arr,
                                  B, S, H, I, nh, float(p_hidden), float(p_attn), float(mods[4].variance_epsilon),
                                  seed, sid, stream_ptr()), "vb_bert_layer_fwd")
        ctx.layer = layer
        ctx.cfg = (B, S, H, I, nh, p_hidden, p_attn, seed, sid)
        ctx.save_for_backward(h2, mask_add, saved, out)     # out: the bf16 output LayerNorm's backward may rebuild x-hat from it
        return out.view(B, S, H)

    @staticmethod
    def backward(ctx, dy):
        h2, mask_add, saved, out = ctx.saved_tensors
        B, S, H, I, nh, p_hidden, p_attn, seed, sid = ctx.cfg
        layer = ctx.layer
        pc = layer_params(layer)
        (at, sa, so, im, om), _, _ = pc
        dt = h2.dtype
        code = _lib.VB_BF16X3 if (_x3[0] and dt == torch.float32) else _lib.dtype_code(dt)
        dy2 = dy.reshape(B * S, H)
        if dy2.dtype != dt:
            dy2 = dy2.to(dt)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        L = _lib.lib()
        scratch = layer_scratch(_layer_bytes(L, code, B, S, H, I, nh, p_attn)[1], h2.device)
        plan = _layer_plan(layer, pc, dt)
        if plan is not None:
            d_in = torch.empty((B * S, H), dtype=dt, device=h2.device)
            plan.arena.touched.update(plan.touched)          # "this step wrote a gradient for these parameters" (grad_target's bookkeeping)
            check(L.vb_bert_layer_bwd(code, ptr(h2), ptr(out), ptr(mask_add), ptr(dy2), ptr(d_in), ptr(saved), ptr(scratch),
                                      plan.w_arr, plan.g_arr, plan.wt_arr, plan.ld_arr, B, S, H, I, nh,
                                      float(p_hidden), float(p_attn), seed, sid, stream_ptr()), "vb_bert_layer_bwd")
            return (d_in.view(B, S, H), None, None, None, None) + _NONE16
        weights = [weight_for(sa.qkv_weight, dt), sa.qkv_bias, weight_for(so.dense.weight, dt), so.dense.bias.detach(),
                   so.LayerNorm.weight.detach(), so.LayerNorm.bias.detach(),
                   weight_for(im.dense.weight, dt), im.dense.bias.detach(),
                   weight_for(om.dense.weight, dt), om.dense.bias.detach(),
                   om.LayerNorm.weight.detach(), om.LayerNorm.bias.detach()]
        g_qkv_w, g_qkv_b, direct_qkv = sa.qkv_grad_targets()
        rest = [so.dense.weight, so.dense.bias, so.LayerNorm.weight, so.LayerNorm.bias, im.dense.weight, im.dense.bias,
                om.dense.weight, om.dense.bias, om.LayerNorm.weight, om.LayerNorm.bias]
        tg = [grad_target(p) for p in rest]
        grads = [g_qkv_w, g_qkv_b] + [t[0] for t in tg]
        d_in = torch.empty((B * S, H), dtype=dt, device=h2.device)
        import ctypes
        wts = [weight_t_for(sa.qkv_weight, dt), weight_t_for(so.dense.weight, dt), weight_t_for(im.dense.weight, dt),
               weight_t_for(om.dense.weight, dt)]
        wt_arr = (ctypes.c_void_p * 4)()
        ld_arr = (ctypes.c_int64 * 4)()
        for i, w_ in enumerate(wts):
            if isinstance(w_, SplitOperand):
                w_ = w_.buf
            wt_arr[i] = w_.data_ptr() if w_ is not None else None
            ld_arr[i] = w_.stride(0) if w_ is not None else 0
        check(L.vb_bert_layer_bwd(code, ptr(h2), ptr(out), ptr(mask_add), ptr(dy2), ptr(d_in), ptr(saved), ptr(scratch),
                                  _ptr_array(weights), _ptr_array(grads), wt_arr, ld_arr, B, S, H, I, nh,
                                  float(p_hidden), float(p_attn), seed, sid, stream_ptr()), "vb_bert_layer_bwd")
        if direct_qkv:
            gq = [None] * 6
        else:
            gw, gb = g_qkv_w.view(3, H, H), g_qkv_b.view(3, H)
            gq = [gw[0], gb[0], gw[1], gb[1], gw[2], gb[2]]
        return (d_in.view(B, S, H), None, None, None, None, *gq, *[grad_result(g, d) for g, d in tg])


@x3_aware
class EmbeddingsFn(torch.autograd.Function):
    """BertEmbeddingsWithVisualEmbedding.forward (modeling.py:1198-1257): region projection GEMM, gather-add of
    the five tables (+ the mean text-position embedding of the aligned words when image_text_alignment is given,
    :1223-1245), concat, LayerNorm, dropout."""

    @staticmethod
    def forward(ctx, module, input_ids, token_type_ids, visual_embeddings, visual_type, alignment, dtype, p_hidden,
                sid, *params):
        m = module
        B, T = input_ids.shape
        dev = input_ids.device
        H = m.word_embeddings.weight.size(1)
        R = 0 if visual_embeddings is None else visual_embeddings.size(1)
        feats = vp = None
        if R > 0:
            Dv = visual_embeddings.size(2)
            f2 = visual_embeddings.reshape(B * R, Dv)
            if f2.dtype != dtype or not f2.is_contiguous() or (Dv % 8) != 0:
                feats = alloc2d(B * R, Dv, dtype, dev, zero=(Dv % 8) != 0)
                if f2.dtype == torch.float32 and f2.is_contiguous() and (Dv % 8) == 0:
                    cast(f2, feats)
                else:
                    feats.copy_(f2)
            else:
                feats = f2
            vp = linear_fwd(feats, weight_for(m.projection.weight, dtype), m.projection.bias.detach())
        pos_align = al = None
        if R > 0 and alignment is not None:
            al = alignment.contiguous()
            if al.dim() != 3 or al.size(0) != B or al.size(1) < R:             # modeling.py:1241-1243
                raise AssertionError("image_text_alignment must be [batch, >= regions, alignment_number]")
            pos_align = torch.empty((B * R, H), dtype=torch.float32, device=dev)
            check(_lib.lib().vb_align_pos_fwd(ptr(al), ptr(m.position_embeddings.weight.detach()), ptr(pos_align),
                                              B, R, al.size(1), al.size(2), H, m.position_embeddings.weight.size(0),
                                              stream_ptr()), "vb_align_pos_fwd")
        z = torch.empty((B * (T + R), H), dtype=dtype, device=dev)
        ids = input_ids.contiguous()
        tt = token_type_ids.contiguous() if token_type_ids is not None else None
        vt = visual_type.contiguous() if visual_type is not None else None
        W = m.word_embeddings.weight
        check(_lib.lib().vb_embed_fwd(
            _lib.dtype_code(dtype), ptr(ids), ptr(tt), ptr(vt), ptr(vp), ptr(W.detach()),
            ptr(m.position_embeddings.weight.detach()), ptr(m.token_type_embeddings.weight.detach()),
            ptr(m.position_embeddings_visual.weight.detach()), ptr(m.token_type_embeddings_visual.weight.detach()),
            ptr(pos_align), ptr(z), B, T, R, H, W.size(0), m.token_type_embeddings.weight.size(0),
            m.position_embeddings.weight.size(0), stream_ptr()), "vb_embed_fwd")
        seed = next_seed()
        y, _, mean, rstd = ln_fwd(z, None, m.LayerNorm.weight.detach(), m.LayerNorm.bias.detach(),
                                  m.LayerNorm.variance_epsilon, 0.0, 0, p_hidden, sid, seed, save_z=False)
        ctx.module = m
        ctx.cfg = (B, T, R, H, dtype, p_hidden, sid, seed)
        ctx.ids = (ids, tt, vt, al)
        ctx.save_for_backward(z, mean, rstd, feats)
        return y.view(B, T + R, H)

    @staticmethod
    def backward(ctx, dy):
        z, mean, rstd, feats = ctx.saved_tensors
        m = ctx.module
        B, T, R, H, dtype, p_hidden, sid, seed = ctx.cfg
        ids, tt, vt, al = ctx.ids
        dy2 = dy.reshape(B * (T + R), H)
        if dy2.dtype != dtype:
            dy2 = dy2.to(dtype)
        g_lw, d1 = grad_target(m.LayerNorm.weight)
        g_lb, d2 = grad_target(m.LayerNorm.bias)
        dz, _ = ln_bwd(dy2, z, mean, rstd, m.LayerNorm.weight.detach(), g_lw, g_lb, None, 0.0, 0, p_hidden, sid, seed)
        g_word, d3 = grad_target(m.word_embeddings.weight)
        g_pos, d4 = grad_target(m.position_embeddings.weight)
        g_type, d5 = grad_target(m.token_type_embeddings.weight)
        g_posv = g_typev = None
        d6 = d7 = True
        if R > 0:                                   # text-only input: the visual tables take no part (their .grad stays None upstream)
            g_posv, d6 = grad_target(m.position_embeddings_visual.weight)
            g_typev, d7 = grad_target(m.token_type_embeddings_visual.weight)
        dvp = torch.empty((B * R, H), dtype=dtype, device=dz.device) if R > 0 else None
        W = m.word_embeddings.weight
        check(_lib.lib().vb_embed_bwd(
            _lib.dtype_code(dtype), ptr(dz), ptr(ids), ptr(tt), ptr(vt), ptr(g_word), ptr(g_pos), ptr(g_type),
            ptr(g_posv), ptr(g_typev), ptr(dvp), B, T, R, H, W.size(0), m.token_type_embeddings.weight.size(0),
            m.position_embeddings.weight.size(0), stream_ptr()), "vb_embed_bwd")
        if al is not None:
            check(_lib.lib().vb_align_pos_bwd(_lib.dtype_code(dtype), ptr(dz), ptr(al), ptr(g_pos), B, T, R, al.size(1),
                                              al.size(2), H, m.position_embeddings.weight.size(0), stream_ptr()),
                  "vb_align_pos_bwd")
        g_pw = g_pb = None
        d8 = d9 = True
        if R > 0:
            g_pw, d8 = grad_target(m.projection.weight)
            g_pb, d9 = grad_target(m.projection.bias)
            linear_wgrad(dvp, feats, g_pw)
            colsum(dvp, g_pb)
        return (None, None, None, None, None, None, None, None, None,
                grad_result(g_word, d3), grad_result(g_pos, d4), grad_result(g_type, d5),
                grad_result(g_lw, d1), grad_result(g_lb, d2), grad_result(g_typev, d7), grad_result(g_posv, d6),
                grad_result(g_pw, d8), grad_result(g_pb, d9))


@x3_aware
class MLMHeadLossFn(torch.autograd.Function):
    """BertLMPredictionHead + CrossEntropyLoss(ignore_index=-1), fused (modeling.py:397-401, 417-420,
    1471-1473): transform GEMM+GELU -> LayerNorm -> tied-decoder GEMM (fp32 logits) -> loss and
    d(logits) in one sweep.  Returns (logits [B,S,V] fp32 view, masked_lm_loss)."""

    @staticmethod
    def forward(ctx, seq, labels, head, word_weight, *params):
        B, S, H = seq.shape
        s2 = seq.reshape(B * S, H)
        if not s2.is_contiguous():
            s2 = s2.contiguous()
        dt = s2.dtype
        tr = head.transform
        pre = torch.empty((B * S, H), dtype=dt, device=s2.device)
        t = linear_fwd(s2, weight_for(tr.dense.weight, dt), tr.dense.bias.detach(), VB_ACT_GELU, aux_out=pre)
        tn, _, mean, rstd = ln_fwd(t, None, tr.LayerNorm.weight.detach(), tr.LayerNorm.bias.detach(),
                                   tr.LayerNorm.variance_epsilon, save_z=False)
        E = weight_for(word_weight, dt)
        V = E.size(0)
        logits = linear_fwd(tn, E, head.bias.detach(), out_dtype=torch.float32)
        loss = None
        dlogits = None
        rows = None
        if labels is not None:
            acc = torch.empty(66, dtype=torch.float32, device=s2.device)
            loss = torch.empty(1, dtype=torch.float32, device=s2.device)
            lab = labels.reshape(-1).contiguous()
            # Rows whose label is ignored (~88 % of an MLM batch) have an exactly zero gradient: the backward GEMMs of
            # the decoder run over the masked rows only.  Their number sizes the compact buffers; it comes from the
            # plan made at the start of the forward (MaskedRowPlan), not from a queue-draining sync here.
            rows = _rows_for(lab)
            n = int(rows.numel())
            n_pad = round_up(max(n, 1), 64)
            dlogits = torch.empty((n_pad, round_up(V, 64)), dtype=dt, device=s2.device)[:, :V]
            check(_lib.lib().vb_ce_fwd_bwd_rows(_lib.dtype_code(dt), ptr(logits), _ld(logits), ptr(lab), -1,
                                                ptr(rows) if n else None, n, n_pad, ptr(acc), ptr(loss), ptr(dlogits),
                                                _ld(dlogits), B * S, V, stream_ptr()), "vb_ce_fwd_bwd_rows")
            loss = loss.reshape(())
        ctx.head, ctx.word_weight = head, word_weight
        ctx.cfg = (B, S, H, V)
        ctx.save_for_backward(s2, pre, t, tn, mean, rstd, dlogits, rows)
        out_logits = logits.view(B, S, V) if logits.is_contiguous() else logits.as_strided(
            (B, S, V), (S * logits.stride(0), logits.stride(0), 1), logits.storage_offset())
        ctx.mark_non_differentiable(out_logits)
        ctx.set_materialize_grads(False)      # else autograd zero-fills a [B,S,V] fp32 gradient for the logits (2.6 GB at B=128)
        if loss is None:
            return out_logits, torch.zeros((), device=s2.device)
        return out_logits, loss

    @staticmethod
    def backward(ctx, _dlogits_unused, dloss):
        s2, pre, t, tn, mean, rstd, dlogits, rows = ctx.saved_tensors
        if dlogits is None:
            raise RuntimeError("visualbert_amd: backward through the MLM head needs masked_lm_labels")
        B, S, H, V = ctx.cfg
        head = ctx.head
        tr = head.transform
        dt = s2.dtype
        up = _upstream_scalar(dloss)
        E = weight_for(ctx.word_weight, dt)
        Et = weight_t_for(ctx.word_weight, dt)
        n, n_pad = int(rows.numel()), dlogits.size(0)
        # compact operands: dlogits is [n_pad, V] (rows of the masked tokens, zero padding), tn_c the same rows of tn
        tn_c = torch.zeros((n_pad, H), dtype=dt, device=tn.device)
        if n:
            tn_c[:n] = tn.index_select(0, rows)
        # dgrad over few rows and a 30522-long reduction: split-K (fp32 accumulate) instead of 30 output tiles
        dtn_c = torch.zeros((n_pad, H), dtype=torch.float32, device=tn.device)
        k_pad = None
        et_ld = (Et.ld // 2 if isinstance(Et, SplitOperand) else Et.stride(0)) if Et is not None else 0
        if Et is not None and et_ld == _ld(dlogits) and (et_ld % 64) == 0:
            k_pad = et_ld                           # both pads are zero: reduce over whole K tiles (LDS-direct)
        if isinstance(Et, SplitOperand) and (k_pad or V) % 64 != 0:
            Et = None                               # ragged reduction: the exact fp32 kernels on the master weight
        if Et is not None:
            gemm(dlogits, Et, n_pad, H, k_pad or V, out=dtn_c, accumulate=True, alpha_dev=up)
        else:
            gemm(dlogits, E, n_pad, H, V, b_layout=VB_KSTRIDED, out=dtn_c, accumulate=True, alpha_dev=up)
        dtn = torch.zeros((B * S, H), dtype=dt, device=tn.device)
        if n:
            dtn.index_copy_(0, rows, dtn_c[:n].to(dt))
        g_E, d1 = grad_target(ctx.word_weight)
        linear_wgrad(dlogits, tn_c, g_E, alpha_dev=up)
        g_db, d2 = grad_target(head.bias)
        colsum(dlogits, g_db, scale_dev=up)
        g_lw, d3 = grad_target(tr.LayerNorm.weight)
        g_lb, d4 = grad_target(tr.LayerNorm.bias)
        dt_, _ = ln_bwd(dtn, t, mean, rstd, tr.LayerNorm.weight.detach(), g_lw, g_lb)
        dpre = act_bwd(dt_, pre, VB_ACT_GELU)
        g_tw, d5 = grad_target(tr.dense.weight)
        g_tb, d6 = grad_target(tr.dense.bias)
        linear_wgrad(dpre, s2, g_tw)
        colsum(dpre, g_tb)
        dseq = linear_dgrad(dpre, weight_for(tr.dense.weight, dt), wt=weight_t_for(tr.dense.weight, dt))
        return (dseq.view(B, S, H), None, None, grad_result(g_E, d1), grad_result(g_db, d2), grad_result(g_tw, d5),
                grad_result(g_tb, d6), grad_result(g_lw, d3), grad_result(g_lb, d4))


@x3_aware
class SparseMLMHeadLossFn(torch.autograd.Function):
    """Opt-in MLM head over the LABELLED positions only: gather the rows whose label is counted, then
    transform GEMM+GELU -> LayerNorm -> tied-decoder GEMM -> CrossEntropyLoss on those rows.  Same loss and gradients as
    MLMHeadLossFn (rows with an ignored label contribute nothing to either), without the ~88 % of the decoder forward and
    the [B,S,V] logits tensor.  Returns (logits [n, V] fp32 of the labelled rows, row indices [n] into B*S, loss).
    pytorch_pretrained_bert/modeling.py:397-401, 417-420, 1466-1473."""

    @staticmethod
    def forward(ctx, seq, labels, head, word_weight, *params):
        B, S, H = seq.shape
        s2 = seq.reshape(B * S, H)
        dt = s2.dtype
        tr = head.transform
        lab = labels.reshape(-1).contiguous()
        rows = _rows_for(lab)
        n = int(rows.numel())
        n_pad = round_up(max(n, 1), 64)
        s_c = torch.zeros((n_pad, H), dtype=dt, device=s2.device)
        lab_c = torch.full((n_pad,), -1, dtype=torch.int64, device=s2.device)
        if n:
            s_c[:n] = s2.index_select(0, rows)
            lab_c[:n] = lab.index_select(0, rows)
        pre = torch.empty((n_pad, H), dtype=dt, device=s2.device)
        t = linear_fwd(s_c, weight_for(tr.dense.weight, dt), tr.dense.bias.detach(), VB_ACT_GELU, aux_out=pre)
        tn, _, mean, rstd = ln_fwd(t, None, tr.LayerNorm.weight.detach(), tr.LayerNorm.bias.detach(),
                                   tr.LayerNorm.variance_epsilon, save_z=False)
        E = weight_for(word_weight, dt)
        V = E.size(0)
        logits = linear_fwd(tn, E, head.bias.detach(), out_dtype=torch.float32)
        acc = torch.empty(66, dtype=torch.float32, device=s2.device)
        loss = torch.empty(1, dtype=torch.float32, device=s2.device)
        dlogits = torch.empty((n_pad, round_up(V, 64)), dtype=dt, device=s2.device)[:, :V]
        check(_lib.lib().vb_ce_fwd_bwd(_lib.dtype_code(dt), ptr(logits), _ld(logits), ptr(lab_c), -1, ptr(acc), ptr(loss),
                                       ptr(dlogits), _ld(dlogits), n_pad, V, stream_ptr()), "vb_ce_fwd_bwd")
        ctx.head, ctx.word_weight = head, word_weight
        ctx.cfg = (B, S, H, V, n)
        ctx.save_for_backward(s_c, pre, t, tn, mean, rstd, dlogits, rows)
        out_logits = logits[:n]
        ctx.mark_non_differentiable(out_logits, rows)
        ctx.set_materialize_grads(False)
        return out_logits, rows, loss.reshape(())

    @staticmethod
    def backward(ctx, _dl, _dr, dloss):
        s_c, pre, t, tn, mean, rstd, dlogits, rows = ctx.saved_tensors
        B, S, H, V, n = ctx.cfg
        head = ctx.head
        tr = head.transform
        dt = s_c.dtype
        n_pad = s_c.size(0)
        up = _upstream_scalar(dloss)
        E = weight_for(ctx.word_weight, dt)
        Et = weight_t_for(ctx.word_weight, dt)
        dtn32 = torch.zeros((n_pad, H), dtype=torch.float32, device=s_c.device)
        k_pad = None
        et_ld = (Et.ld // 2 if isinstance(Et, SplitOperand) else Et.stride(0)) if Et is not None else 0
        if Et is not None and et_ld == _ld(dlogits) and (et_ld % 64) == 0:
            k_pad = et_ld
        if isinstance(Et, SplitOperand) and (k_pad or V) % 64 != 0:
            Et = None                               # ragged reduction: the exact fp32 kernels on the master weight
        if Et is not None:
            gemm(dlogits, Et, n_pad, H, k_pad or V, out=dtn32, accumulate=True, alpha_dev=up)
        else:
            gemm(dlogits, E, n_pad, H, V, b_layout=VB_KSTRIDED, out=dtn32, accumulate=True, alpha_dev=up)
        dtn = dtn32.to(dt)
        g_E, d1 = grad_target(ctx.word_weight)
        linear_wgrad(dlogits, tn, g_E, alpha_dev=up)
        g_db, d2 = grad_target(head.bias)
        colsum(dlogits, g_db, scale_dev=up)
        g_lw, d3 = grad_target(tr.LayerNorm.weight)
        g_lb, d4 = grad_target(tr.LayerNorm.bias)
        dt_, _ = ln_bwd(dtn, t, mean, rstd, tr.LayerNorm.weight.detach(), g_lw, g_lb)
        dpre = act_bwd(dt_, pre, VB_ACT_GELU)
        g_tw, d5 = grad_target(tr.dense.weight)
        g_tb, d6 = grad_target(tr.dense.bias)
        linear_wgrad(dpre, s_c, g_tw)
        colsum(dpre, g_tb)
        dseq_c = linear_dgrad(dpre, weight_for(tr.dense.weight, dt), wt=weight_t_for(tr.dense.weight, dt))
        dseq = torch.zeros((B * S, H), dtype=dt, device=s_c.device)
        if n:
            dseq.index_copy_(0, rows, dseq_c[:n])
        return (dseq.view(B, S, H), None, None, grad_result(g_E, d1), grad_result(g_db, d2), grad_result(g_tw, d5),
                grad_result(g_tb, d6), grad_result(g_lw, d3), grad_result(g_lb, d4))


@x3_aware
class SmallLinearCEFn(torch.autograd.Function):
    """Linear with a tiny output width + CrossEntropyLoss: seq_relationship / image-text-match
    (modeling.py:451, 1474), the NLVR2 classifier (modeling.py:1558-1565) and, with choices = 4, the VCR
    multiple-choice head (modeling.py:1488-1500: Linear H -> 1 per choice, the loss over logits.view(-1, 4)).
    Returns (logits fp32 [M / choices, N * choices], loss)."""

    @staticmethod
    def forward(ctx, x, labels, ignore_index, weight, bias, choices=1):
        x2 = as2d(x)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M, K = x2.shape
        N = weight.size(0)
        y = torch.empty((M, N), dtype=torch.float32, device=x2.device)
        check(_lib.lib().vb_small_linear_fwd(_lib.dtype_code(x2.dtype), ptr(x2), K, ptr(weight.detach()),
                                             ptr(bias.detach()), ptr(y), M, N, K, stream_ptr()), "vb_small_linear_fwd")
        loss = torch.zeros((), dtype=torch.float32, device=x2.device)
        dy = None
        if labels is not None:
            acc = torch.empty(66, dtype=torch.float32, device=x2.device)
            l1 = torch.empty(1, dtype=torch.float32, device=x2.device)
            dy = torch.empty((M, N), dtype=torch.float32, device=x2.device)
            if M % choices != 0:
                raise RuntimeError("visualbert_amd: %d rows do not divide into groups of %d choices" % (M, choices))
            cm, cv = M // choices, N * choices            # the loss sees y as [M / choices, N * choices] (same memory)
            lab_c = labels.reshape(-1).contiguous()
            check(_lib.lib().vb_ce_fwd_bwd(_lib.VB_F32, ptr(y), cv, ptr(lab_c), ignore_index,
                                           ptr(acc), ptr(l1), ptr(dy), cv, cm, cv, stream_ptr()), "vb_ce_fwd_bwd")
            loss = l1.reshape(())
        ctx.wb = (weight, bias)
        ctx.x_shape = x.shape
        ctx.save_for_backward(x2, dy)
        if choices != 1:
            y = y.view(M // choices, N * choices)
        ctx.mark_non_differentiable(y)
        ctx.set_materialize_grads(False)
        return y, loss

    @staticmethod
    def backward(ctx, _dy_unused, dloss):
        x2, dy = ctx.saved_tensors
        if dy is None:
            raise RuntimeError("visualbert_amd: backward through this head needs labels")
        weight, bias = ctx.wb
        M, K = x2.shape
        N = weight.size(0)
        up = _upstream_scalar(dloss)
        gw, d1 = grad_target(weight)
        gb, d2 = grad_target(bias)
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        check(_lib.lib().vb_small_linear_bwd(_lib.dtype_code(x2.dtype), ptr(dy), ptr(x2), K, ptr(weight.detach()),
                                             ptr(dx), K, ptr(gw), ptr(gb), ptr(up), M, N, K, stream_ptr()),
              "vb_small_linear_bwd")
        return (dx.view(ctx.x_shape) if dx is not None else None, None, None, grad_result(gw, d1), grad_result(gb, d2),
                None)


@x3_aware
class VQAHeadLossFn(torch.autograd.Function):
    """VQA head (modeling.py:1502-1525): gather hidden state at index input_mask.sum(1)-2, Linear H->3129,
    KLDivLoss(batchmean) on log_softmax, mean VQA score.  Returns (logits [B,1,3129] fp32, loss, accuracy)."""

    @staticmethod
    def forward(ctx, seq, input_mask, target, p_drop, sid, weight, bias):
        B, S, H = seq.shape
        s2 = seq.reshape(B * S, H)
        if not s2.is_contiguous():
            s2 = s2.contiguous()
        dt = s2.dtype
        T = input_mask.size(1)
        g = torch.empty((B, H), dtype=dt, device=s2.device)
        idx = torch.empty(B, dtype=torch.int64, device=s2.device)
        im_c = input_mask.contiguous()
        check(_lib.lib().vb_gather_rows(_lib.dtype_code(dt), ptr(s2), ptr(im_c), ptr(g), ptr(idx),
                                        B, S, T, H, stream_ptr()), "vb_gather_rows")
        seed = next_seed()
        if p_drop > 0.0:
            g = dropout_apply(g, p_drop, seed, sid)         # nn.Dropout on the gathered state (modeling.py:1509); regenerated in backward
        N = weight.size(0)
        logits = linear_fwd(g, weight_for(weight, dt), bias.detach(), out_dtype=torch.float32)
        loss = torch.zeros(1, dtype=torch.float32, device=s2.device)
        score = torch.zeros(1, dtype=torch.float32, device=s2.device)
        dlogits = None
        if target is not None:
            dlogits = alloc2d(B, N, torch.float32, s2.device, zero=True)
            tg = target.contiguous().to(torch.float32)
            check(_lib.lib().vb_kldiv_fwd_bwd(ptr(logits), _ld(logits), ptr(tg), tg.stride(0), ptr(loss), ptr(score),
                                              ptr(dlogits), _ld(dlogits), B, N, stream_ptr()), "vb_kldiv_fwd_bwd")
        ctx.wb = (weight, bias)
        ctx.cfg = (B, S, H, N, p_drop, seed, sid)
        ctx.save_for_backward(g, idx, dlogits)
        out = logits.contiguous().view(B, 1, N)
        ctx.mark_non_differentiable(out, idx)
        ctx.set_materialize_grads(False)
        return out, loss.reshape(()), score.reshape(()), idx

    @staticmethod
    def backward(ctx, _dl, dloss, _ds, _di):
        g, idx, dlogits = ctx.saved_tensors
        if dlogits is None:
            raise RuntimeError("visualbert_amd: backward through the VQA head needs labels")
        weight, bias = ctx.wb
        B, S, H, N, p_drop, seed, sid = ctx.cfg
        dt = g.dtype
        up = _upstream_scalar(dloss)
        dl = alloc2d(B, N, dt, g.device, zero=True)
        dl.copy_(dlogits if up is None else dlogits * up)
        gw, d1 = grad_target(weight)
        gb, d2 = grad_target(bias)
        linear_wgrad(dl, g, gw)
        colsum(dl, gb)
        dg = linear_dgrad(dl, weight_for(weight, dt))
        if p_drop > 0.0:
            dg = dropout_apply(dg, p_drop, seed, sid)       # the forward's mask, regenerated from (seed, sid)
        dseq = torch.zeros((B * S, H), dtype=dt, device=g.device)
        dg = dg.contiguous()
        check(_lib.lib().vb_scatter_rows(_lib.dtype_code(dt), ptr(dg), ptr(idx), ptr(dseq), B, S, H,
                                         stream_ptr()), "vb_scatter_rows")
        return dseq.view(B, S, H), None, None, None, None, grad_result(gw, d1), grad_result(gb, d2)


@x3_aware
class FlickrHeadLossFn(torch.autograd.Function):
    """Flickr30k grounding head (modeling.py:1568-1598): batched_index_select of the entity positions, the query / key
    projections of FlickrAttention (:1624-1648; one head, no value, no softmax), masked scores over the regions,
    KLDivLoss(batchmean) on their log-softmax, compute_score_with_logits_flickr (:1650-1673).
    Returns (loss, hits, label mass, entities_num) -- the caller forms accuracy = hits / entities_num."""

    @staticmethod
    def forward(ctx, seq, position, image_mask, label, T, d, wq, bq, wk, bk):
        B, S, H = seq.shape
        R = S - T
        E = position.size(1)
        s2 = seq.reshape(B * S, H)
        if not s2.is_contiguous():
            s2 = s2.contiguous()
        dt, dev = s2.dtype, s2.device
        L = _lib.lib()
        pos = position.contiguous()
        sel = torch.empty((B * E, H), dtype=dt, device=dev)
        check(L.vb_gather_index_rows(_lib.dtype_code(dt), ptr(s2), ptr(pos), ptr(sel), B, S, E, H, stream_ptr()),
              "vb_gather_index_rows")
        q = linear_fwd(sel, weight_for(wq, dt), bq.detach())
        k = linear_fwd(s2, weight_for(wk, dt), bk.detach())        # keys of every position; the regions are rows T..S-1
        scores = torch.empty((B * E, R), dtype=torch.float32, device=dev)
        stats = torch.empty(3, dtype=torch.float32, device=dev)
        lab = label.contiguous().to(torch.float32).view(B * E, R)
        im = image_mask.contiguous()
        check(L.vb_flickr_scores_fwd(_lib.dtype_code(dt), ptr(q), _ld(q), ptr(k), _ld(k), ptr(im), ptr(lab), ptr(pos),
                                     ptr(scores), ptr(stats), B, E, R, S, T, d, stream_ptr()), "vb_flickr_scores_fwd")
        loss = torch.zeros(1, dtype=torch.float32, device=dev)
        ds = torch.empty((B * E, R), dtype=torch.float32, device=dev)
        check(L.vb_kldiv_fwd_bwd(ptr(scores), R, ptr(lab), R, ptr(loss), None, ptr(ds), R, B * E, R, stream_ptr()),
              "vb_kldiv_fwd_bwd")
        # the kernel averages over its B*E rows; KLDivLoss(batchmean) on a [B, E, R] input divides by B
        ctx.wb = (wq, bq, wk, bk)
        ctx.cfg = (B, S, H, T, R, E, d)
        ctx.save_for_backward(s2, pos, sel, q, k, ds)
        ctx.set_materialize_grads(False)
        hits, upper, n_ent = stats[0], stats[1], stats[2]
        ctx.mark_non_differentiable(hits, upper, n_ent)
        return loss.reshape(()) * float(E), hits, upper, n_ent

    @staticmethod
    def backward(ctx, dloss, _a, _b, _c):
        s2, pos, sel, q, k, ds = ctx.saved_tensors
        wq, bq, wk, bk = ctx.wb
        B, S, H, T, R, E, d = ctx.cfg
        dt = s2.dtype
        L = _lib.lib()
        up = _upstream_scalar(dloss)
        dq = torch.empty_like(q)
        dk = torch.empty_like(k)
        check(L.vb_flickr_scores_bwd(_lib.dtype_code(dt), ptr(ds), ptr(q), _ld(q), ptr(k), _ld(k), ptr(dq), ptr(dk),
                                     ptr(up), float(E), B, E, R, S, T, d, stream_ptr()), "vb_flickr_scores_bwd")
        gwq, d1 = grad_target(wq)
        gbq, d2 = grad_target(bq)
        gwk, d3 = grad_target(wk)
        gbk, d4 = grad_target(bk)
        linear_wgrad(dq, sel, gwq)
        colsum(dq, gbq)
        linear_wgrad(dk, s2, gwk)
        colsum(dk, gbk)
        dsel = linear_dgrad(dq, weight_for(wq, dt))
        dseq_k = linear_dgrad(dk, weight_for(wk, dt))
        dseq = torch.empty((B * S, H), dtype=dt, device=s2.device)
        dsel, dseq_k = dsel.contiguous(), dseq_k.contiguous()
        check(L.vb_scatter_index_rows(_lib.dtype_code(dt), ptr(dsel), ptr(pos), ptr(dseq_k),
                                      ptr(dseq), B, S, E, H, stream_ptr()), "vb_scatter_index_rows")
        return (dseq.view(B, S, H), None, None, None, None, None, grad_result(gwq, d1), grad_result(gbq, d2),
                grad_result(gwk, d3), grad_result(gbk, d4))
