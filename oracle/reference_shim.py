"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (visualbert_amd/).

Imports the reference's own hot-path modules from /root/reference (read-only,
exists only in the build container, NOT on the GPU box) so that
  * oracle/visualbert_oracle.py (the restatement that travels) can be pinned
    against the real reference, and
  * oracle/make_golden.py can dump golden vectors into tests/golden/.

Two harness-level shims are needed:
  1. `boto3` / `botocore` are absent -> stub modules in sys.modules
     (reference pytorch_pretrained_bert/file_utils.py:20-21 imports them).
  2. modeling.py:1238,1247 call `.cuda()` unconditionally -> while a reference
     forward runs, torch.Tensor.cuda is replaced by the identity.
Nothing from the reference is copied; it is imported in place.
"""
import contextlib
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("VB_REFERENCE_ROOT", "/root/reference")
_REF_PKG_PARENT = os.path.join(REFERENCE_ROOT, "visualbert")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(_REF_PKG_PARENT, "pytorch_pretrained_bert", "modeling.py"))


def _install_stubs():
    if "boto3" not in sys.modules:
        sys.modules["boto3"] = types.ModuleType("boto3")
    if "botocore" not in sys.modules:
        bc = types.ModuleType("botocore")
        bce = types.ModuleType("botocore.exceptions")

        class ClientError(Exception):
            pass

        bce.ClientError = ClientError
        bc.exceptions = bce
        sys.modules["botocore"] = bc
        sys.modules["botocore.exceptions"] = bce


_cached = None


def load_reference():
    """Returns (modeling, optimization) modules of the reference."""
    global _cached
    if _cached is not None:
        return _cached
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    _install_stubs()
    if _REF_PKG_PARENT not in sys.path:
        sys.path.insert(0, _REF_PKG_PARENT)
    import io
    with contextlib.redirect_stdout(io.StringIO()):  # silences the apex hint print
        from pytorch_pretrained_bert import modeling as ref_modeling
        from pytorch_pretrained_bert import optimization as ref_optimization
    _cached = (ref_modeling, ref_optimization)
    return _cached


@contextlib.contextmanager
def cpu_cuda_noop():
    """Neutralises Tensor.cuda() for the duration of a reference call on CPU."""
    import torch
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        yield
    finally:
        torch.Tensor.cuda = orig
