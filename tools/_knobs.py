"""shared by the measurement scripts: route the package through the DEVELOPER build of the library (same kernels + the
knobs of include/visualbert_hip_dev.h) and keep the old knob vocabulary on top of per-stream options."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from visualbert_amd import _lib  # noqa: E402

L = _lib.use_dev_library()
_state = dict(nt_kernel=0, persistent_workgroups=0, attn_two_pass=0)


def _apply():
    _lib.set_opts(**_state)


def variant(v):
    """round-1 numbering: 1 = chosen from the shape, 0 = generic kernel, else the kernel id (22, 42, 80, 81, 90)."""
    _state["nt_kernel"] = 0 if v == 1 else (1 if v == 0 else int(v))
    _apply()
    return 0


def wgs(n):
    _state["persistent_workgroups"] = int(n)
    _apply()
    return 0


def two_pass(on):
    _state["attn_two_pass"] = int(on)
    _apply()
    return 0
