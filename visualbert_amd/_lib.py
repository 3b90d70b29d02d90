"""ctypes binding of libvisualbert_hip.so (the C ABI declared in include/visualbert_hip.h).

There is NO fallback: if the gfx950 library is missing or an entry point returns non-zero, a
RuntimeError is raised.  Build it with `python -c "import __graft_entry__ as g; g.build()"` or
`make -C visualbert_amd/csrc`.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "libvisualbert_hip.so")

VB_F32, VB_BF16, VB_BF16X3 = 0, 1, 2
VB_KCONTIG, VB_KSTRIDED = 0, 1
VB_ACT_NONE, VB_ACT_GELU, VB_ACT_TANH, VB_ACT_GELU_GRAD, VB_ACT_GELU_SAVE_GRAD, VB_ACT_MUL_AUX = 0, 1, 2, 3, 4, 5

_i, _i64, _f, _p = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p
_u32, _u64 = ctypes.c_uint32, ctypes.c_uint64

# name -> (restype, argtypes); kept in the order of include/visualbert_hip.h
SIGNATURES = {
    "vb_version": (ctypes.c_char_p, []),
    "vb_stream_set_opts": (_i, [_p, _p]),
    "vb_stream_get_opts": (_i, [_p, _p]),
    "vb_stream_set_scratch": (_i, [_p, _p, _i64]),
    "vb_gemm": (_i, [_i, _i, _i, _i, _p, _i64, _p, _i64, _p, _i64, _i, _i, _i, _f, _p, _p, _p, _i64, _i,
                     _p, _p, _i64, _i, _p, _p]),
    "vb_ln_fwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _f, _u32, _f, _u32, _u64, _p]),
    "vb_ln_bwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _u32, _f, _u32, _u64, _p, _p]),
    "vb_ln_bwd_ws_bytes": (_i64, [_i, _i]),
    "vb_ln_fwd_rb": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _f, _u32, _u64, _p, _p]),
    "vb_ln_bwd_rb": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _f, _u32, _u64, _p, _p, _p, _p, _p]),
    "vb_embed_fwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "vb_attn_probs": (_i, [_i, _p, _p, _p, _i, _i, _i, _i, _p]),
    "vb_align_pos_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "vb_align_pos_bwd": (_i, [_i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "vb_embed_bwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "vb_attn_keepbits_words": (_i64, [_i]),
    "vb_attn_fwd": (_i, [_i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _u64, _u32, _p]),
    "vb_attn_bwd_ws_floats": (_i64, [_i, _i, _i]),
    "vb_attn_cross_keepbits_words": (_i64, [_i, _i]),
    "vb_attn_cross_fwd": (_i, [_i, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _p, _p, _i, _i, _i, _i, _i, _f, _u64, _u32, _p]),
    "vb_attn_cross_bwd": (_i, [_i, _p, _i64, _p, _i64, _p, _i64, _p, _p, _i64, _p, _p, _p, _p, _i64, _p, _i64, _p, _i64,
                               _i, _i, _i, _i, _i, _f, _u64, _u32, _p]),
    "vb_attn_bwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _u64, _u32, _p]),
    "vb_ce_fwd_bwd": (_i, [_i, _p, _i64, _p, _i, _p, _p, _p, _i64, _i, _i, _p]),
    "vb_ce_fwd_bwd_rows": (_i, [_i, _p, _i64, _p, _i, _p, _i, _i, _p, _p, _p, _i64, _i, _i, _p]),
    "vb_kldiv_fwd_bwd": (_i, [_p, _i64, _p, _i64, _p, _p, _p, _i64, _i, _i, _p]),
    "vb_small_linear_fwd": (_i, [_i, _p, _i64, _p, _p, _p, _i, _i, _i, _p]),
    "vb_small_linear_bwd": (_i, [_i, _p, _p, _i64, _p, _p, _i64, _p, _p, _p, _i, _i, _i, _p]),
    "vb_bert_adam_step": (_i, [_p, _p, _p, _p, _p, _p, _i, _p, _i, _p, _p, _p, _f, _f, _f, _f, _f, _f, _f, _f, _i, _p]),
    "vb_refresh_bf16_shadow": (_i, [_p, _p, _p, _i, _p, _p]),
    "vb_refresh_transposed_shadow": (_i, [_p, _p, _p, _p, _i, _p]),
    "vb_prepare_inputs": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p]),
    "vb_zero": (_i, [_p, _i64, _p]),
    "vb_cast": (_i, [_i, _p, _i, _p, _i64, _p]),
    "vb_split_bf16": (_i, [_p, _i64, _p, _i64, _i64, _i, _p]),
    "vb_split_bf16_t": (_i, [_p, _i64, _p, _i64, _i, _i, _p]),
    "vb_dropout": (_i, [_i, _p, _p, _i64, _f, _u64, _u32, _p]),
    "vb_gather_rows": (_i, [_i, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "vb_scatter_rows": (_i, [_i, _p, _p, _p, _i, _i, _i, _p]),
    "vb_gather_index_rows": (_i, [_i, _p, _p, _p, _i, _i, _i, _i, _p]),
    "vb_scatter_index_rows": (_i, [_i, _p, _p, _p, _p, _i, _i, _i, _i, _p]),
    "vb_flickr_scores_fwd": (_i, [_i, _p, _i64, _p, _i64, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "vb_flickr_scores_bwd": (_i, [_i, _p, _p, _i64, _p, _i64, _p, _p, _p, _f, _i, _i, _i, _i, _i, _i, _p]),
    "vb_colsum": (_i, [_i, _p, _i64, _p, _p, _i, _i, _p]),
    "vb_act_bwd": (_i, [_i, _p, _p, _p, _i64, _i, _p]),
    "vb_wgrad_grouped": (_i, [_i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _f, _p, _p]),
    "vb_stream_profile": (_i, [_p, _i]),
    "vb_stream_profile_read": (_i64, [_p, _p, _p, _p, _i64]),
    "vb_bert_layer_saved_bytes": (_i64, [_i, _i, _i, _i, _i, _i, _f]),
    "vb_bert_layer_scratch_bytes": (_i64, [_i, _i, _i, _i, _i, _i]),
    "vb_bert_layer_fwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _f, _u64, _u32, _p]),
    "vb_bert_layer_bwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _u64, _u32, _p]),
    "vb_comm_unique_id": (_i, [_p]),
    "vb_comm_init": (_i, [_p, _i, _i, _p]),
    "vb_comm_nranks": (_i, [_p]),
    "vb_allreduce_bucket": (_i, [_p, _p, _i64, _i, _i, _p]),
    "vb_comm_destroy": (_i, [_p]),
}
# developer build only (include/visualbert_hip_dev.h, libvisualbert_hip_dev.so): never needed by the package
DEV_SIGNATURES = {
    "vb_gemm_set_debug": (_i, [_i]),
    "vb_gemm_set_trace": (_i, [_p]),
    "vb_mfma_peak": (_i, [_i, _i, _i, _p, _p]),
    "vb_glds_stream": (_i, [_i, _p, _i64, _i, _i, _p, _p]),
    "vb_mma_f8_probe": (_i, [_p, _p, _p, _p, _p, _p]),
    "vb_cvt_fp8_probe": (_i, [_p, _p, _i, _p]),
    "vb_split_f8": (_i, [_p, _i64, _p, _i64, _i, _i, _p, _p, _p]),
    "vb_gemm_x3f8": (_i, [_p, _i64, _p, _i64, _p, _i64, _i, _i, _i, _p, _p, _p, _p, _p, _p]),
}
VB_COMM_ID_BYTES = 128


class StreamOpts(ctypes.Structure):
    """include/visualbert_hip.h: vb_stream_opts"""
    _fields_ = [("persistent_workgroups", _i), ("nt_kernel", _i), ("attn_two_pass", _i), ("reserved", _i)]

_ERRORS = {-1: "VB_ERR_ARG (bad argument)", -2: "VB_ERR_LAUNCH (hip launch failed)",
           -3: "VB_ERR_UNSUPPORTED (shape/dtype not supported by this kernel)"}

_lib = None
_lib_path = None
_device_type = "cuda"


def set_library(path, device_type="cuda"):
    """Select the shared object to bind.  The product never calls this (DEFAULT_LIB on `cuda`);
    tests/conftest.py uses it to point the same Python code at the developer-only kernel-logic
    simulator (tests/hipemu) when VB_EMU=1."""
    global _lib, _lib_path, _device_type
    _lib, _lib_path, _device_type = None, path, device_type


def device_type():
    return _device_type


def lib():
    global _lib, _lib_path
    if _lib is not None:
        return _lib
    path = _lib_path or DEFAULT_LIB
    if not os.path.isfile(path):
        raise RuntimeError(
            "visualbert_amd: %s not found -- the HIP extension is required (no CPU fallback exists). "
            "Build it: python -c 'import __graft_entry__ as g; g.build()'" % path)
    L = ctypes.CDLL(path)
    _bind(L, SIGNATURES, path)
    _lib, _lib_path = L, path
    return L


def _bind(L, table, path):
    for name, (res, args) in table.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            raise RuntimeError("visualbert_amd: %s does not export %s (stale build?)" % (path, name))
        fn.restype = res
        fn.argtypes = args


_dev = None


def dev_lib(required=False):
    """libvisualbert_hip_dev.so (the product's objects + the developer knobs of include/visualbert_hip_dev.h) or None.
    Tools that use it bind it INSTEAD of the product library: call `use_dev_library()` before anything else."""
    global _dev
    if _dev is not None:
        return _dev
    path = os.path.join(_HERE, "libvisualbert_hip_dev.so")
    if not os.path.isfile(path):
        if required:
            raise RuntimeError("visualbert_amd: %s not found (make -C visualbert_amd/csrc)" % path)
        return None
    L = ctypes.CDLL(path)
    _bind(L, SIGNATURES, path)
    _bind(L, DEV_SIGNATURES, path)
    _dev = L
    return L


def use_dev_library():
    """route the whole package through the developer build (tools/ only): same kernels plus the dev knobs."""
    global _lib, _lib_path
    L = dev_lib(required=True)
    _lib, _lib_path = L, os.path.join(_HERE, "libvisualbert_hip_dev.so")
    return L


#: vb_stream_opts.nt_kernel values that exist in the developer library only (include/visualbert_hip_dev.h)
DEV_NT_KERNELS = (80, 82, 91, 92, 100, 101, 200)


class dev_library(object):
    """with dev_library(): ...   -- bind libvisualbert_hip_dev.so for the block (tests of the experiment arms, bench.py's
    yardstick legs), then return to whatever library was bound before.  Per-stream options live inside each library: set them
    inside the block."""

    def __enter__(self):
        global _lib, _lib_path
        self.saved = (_lib, _lib_path)
        return use_dev_library()

    def __exit__(self, *exc):
        global _lib, _lib_path
        _lib, _lib_path = self.saved
        return False


class stream_opts(object):
    """with stream_opts(nt_kernel=42, persistent_workgroups=8): ...   -- launch options of the CURRENT stream
    (vb_stream_set_opts); restored on exit.  They never change results beyond summation order."""

    def __init__(self, persistent_workgroups=None, nt_kernel=None, attn_two_pass=None, stream=None):
        self.kw = dict(persistent_workgroups=persistent_workgroups, nt_kernel=nt_kernel, attn_two_pass=attn_two_pass)
        self.stream = stream

    def _sp(self):
        if self.stream is not None:
            return ctypes.c_void_p(self.stream.cuda_stream)
        return stream_ptr()

    def __enter__(self):
        L = lib()
        self.old = StreamOpts()
        check(L.vb_stream_get_opts(self._sp(), ctypes.byref(self.old)), "vb_stream_get_opts")
        new = StreamOpts(self.old.persistent_workgroups, self.old.nt_kernel, self.old.attn_two_pass, 0)
        for k, v in self.kw.items():
            if v is not None:
                setattr(new, k, int(v))
        check(L.vb_stream_set_opts(self._sp(), ctypes.byref(new)), "vb_stream_set_opts")
        return self

    def __exit__(self, *exc):
        L = lib()
        zero = not (self.old.persistent_workgroups or self.old.nt_kernel or self.old.attn_two_pass)
        check(L.vb_stream_set_opts(self._sp(), None if zero else ctypes.byref(self.old)), "vb_stream_set_opts")
        return False


def set_opts(persistent_workgroups=0, nt_kernel=0, attn_two_pass=0):
    """attach launch options to the CURRENT stream until changed again (tools; library code uses `stream_opts`)."""
    o = StreamOpts(int(persistent_workgroups), int(nt_kernel), int(attn_two_pass), 0)
    zero = not (o.persistent_workgroups or o.nt_kernel or o.attn_two_pass)
    check(lib().vb_stream_set_opts(stream_ptr(), None if zero else ctypes.byref(o)), "vb_stream_set_opts")


def check(rc, what):
    if rc != 0:
        raise RuntimeError("visualbert_amd: %s failed: %s" % (what, _ERRORS.get(rc, "error %d" % rc)))


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Tensors must live on the library's device."""
    if t is None:
        return None
    if t.device.type != _device_type:
        raise RuntimeError("visualbert_amd: tensor on %s, library runs on %s" % (t.device, _device_type))
    return ctypes.c_void_p(t.data_ptr())


#: bytes of the per-stream scratch handed to the library (vb_stream_set_scratch): 16 KB of arrival counters + the fp32 partial tiles of
#: the in-launch split-K GEMMs (at most compute-units x 64 KB = 16 MB at a time on MI355X)
SCRATCH_BYTES = 32 << 20
_scratch = {}


def _register_scratch(key, sp, device):
    buf = torch.empty(SCRATCH_BYTES + 256, dtype=torch.uint8, device=device)
    base = (buf.data_ptr() + 255) & ~255                  # the ABI asks for 256-byte alignment
    check(lib().vb_stream_set_scratch(sp, ctypes.c_void_p(base), SCRATCH_BYTES), "vb_stream_set_scratch")
    _scratch[key] = buf                                   # kept alive with the process: the library holds the raw pointer


def stream_ptr():
    """the CURRENT stream as the ABI's `void* stream`.  Every call into the library passes through here, so this is also where a
    stream gets its scratch buffer (include/visualbert_hip.h: vb_stream_set_scratch) the first time it is used -- per bound library
    (the per-stream tables live inside each shared object) and per stream."""
    if _lib is None:
        lib()                                             # binds the library (and fixes _lib_path) on first use
    if _device_type == "cuda":
        # the raw handle straight from torch's C side: torch.cuda.current_stream() builds a Python Stream object per call (~10 us of
        # the ~30 calls of a training step at small batches, where the step is host-bound)
        idx = torch.cuda.current_device()
        raw = torch._C._cuda_getCurrentRawStream(idx)
        sp = ctypes.c_void_p(raw)
        key = (_lib_path, idx, raw)
        if key not in _scratch:
            _register_scratch(key, sp, torch.device("cuda", idx))
        return sp
    key = (_lib_path, -1, 0)
    if key not in _scratch:
        _register_scratch(key, None, torch.device("cpu"))
    return None


def dtype_code(dt):
    if dt == torch.float32:
        return VB_F32
    if dt == torch.bfloat16:
        return VB_BF16
    raise RuntimeError("visualbert_amd: unsupported dtype %s (fp32 / bf16 only)" % dt)
