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


# ---- the reference's host-side data functions ---------------------------------------------------------------
class _Anything(type):
    """metaclass of the universal placeholder: attribute access, calls, subscripts and subclassing all yield placeholders, so
    `from allennlp.x import Y`, `class Z(Y[K, V])` and `@overrides` at import time of the reference's modules succeed."""

    def __getattr__(cls, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Stub

    def __getitem__(cls, item):
        return _Stub


class _Stub(metaclass=_Anything):
    def __init__(self, *a, **k):
        pass

    def __new__(cls, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k and cls is _Stub:
            return a[0]                       # used as a decorator (@overrides): hand the function back
        return super().__new__(cls)

    def __call__(self, *a, **k):
        return _Stub()

    def __getattr__(self, name):
        return _Stub()


class _StubModule(types.ModuleType):
    __path__ = []                             # a package: `import a.b.c` resolves through sys.modules entries below

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Stub


_ABSENT = ["allennlp", "allennlp.data", "allennlp.data.dataset", "allennlp.data.fields", "allennlp.data.instance",
           "allennlp.data.token_indexers", "allennlp.data.tokenizers", "allennlp.data.vocabulary", "allennlp.nn",
           "allennlp.nn.util", "allennlp.common", "allennlp.common.checks", "allennlp.data.fields.sequence_field",
           "allennlp.data.fields.field", "allennlp.data.tokenizers.token", "allennlp.data.token_indexers.token_indexer",
           "h5py", "overrides", "spacy", "spacy.tokens", "torchvision", "torchvision.datasets", "torchvision.datasets.folder",
           "torchvision.transforms", "matplotlib", "scipy.misc"]


def load_reference_data_utils():
    """-> (fine_tuning module, bert_data_utils module) of the reference, imported in place.  Their import-time dependencies
    that are absent here (allennlp 0.8, h5py, spacy, overrides, torchvision, matplotlib) are replaced by inert placeholder
    modules: the two functions this repo pins -- fine_tuning.random_word (:272-308) and
    InputFeatures.convert_one_example_to_features_pretraining (dataloaders/bert_data_utils.py:168-247) -- are plain Python
    over lists and a tokenizer's vocab dict and touch none of them."""
    load_reference()
    import importlib
    for name in _ABSENT:
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = _StubModule(name)
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        from pytorch_pretrained_bert import fine_tuning as ref_ft
        from dataloaders import bert_data_utils as ref_bdu
    return ref_ft, ref_bdu


def load_reference_lxrt(l_layers=2, x_layers=1, r_layers=0, **arg_overrides):
    """-> the sibling model's module unsupervised_visualbert/src/lxrt/modeling.py, imported in place.  It reads a global
    `args` (src/param.py runs argparse at import time): a stand-in `param` module supplies an args object with the three
    layer counts VisualConfig wants and .get(name, default) for every optional switch (all off unless overridden)."""
    _install_stubs()
    src = os.path.join(REFERENCE_ROOT, "unsupervised_visualbert", "src")
    if not os.path.isfile(os.path.join(src, "lxrt", "modeling.py")):
        raise RuntimeError("reference tree not present at %s" % src)

    class _Args(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

    a = _Args(llayers=l_layers, xlayers=x_layers, rlayers=r_layers)
    a.update(arg_overrides)
    mod = types.ModuleType("param")
    mod.args = a
    sys.modules["param"] = mod
    if src not in sys.path:
        sys.path.insert(0, src)
    import io
    import importlib
    with contextlib.redirect_stdout(io.StringIO()):
        return importlib.import_module("lxrt.modeling")
