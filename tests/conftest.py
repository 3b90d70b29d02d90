import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`gpu` tests run on a GPU, or -- developer aid only -- on the kernel-logic
    simulator when VB_EMU=1 (tests/hipemu; the product never loads it)."""
    import torch
    have = torch.cuda.is_available() or os.environ.get("VB_EMU") == "1"
    if have:
        return
    skip = pytest.mark.skip(reason="no GPU (and VB_EMU != 1)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def _emu_lib_path():
    return os.path.join(ROOT, "tests", "hipemu", "libvisualbert_emu.so")


@pytest.fixture(scope="session")
def dev():
    """Device the kernels run on: 'cuda' on the GPU box; 'cpu' under VB_EMU=1 (kernel-logic simulator)."""
    import torch
    from visualbert_amd import _lib
    if os.environ.get("VB_EMU") == "1":
        path = _emu_lib_path()
        if not os.path.isfile(path):
            pytest.skip("VB_EMU=1 but %s is not built (make -C visualbert_amd/csrc emu)" % path)
        _lib.set_library(path, "cpu")
        return torch.device("cpu")
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)
