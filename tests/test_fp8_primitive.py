"""The block-scaled fp8 MFMA primitive (csrc/vb_rt.h: vb_mma_f8, vb_cvt4_fp8) -- groundwork for running the two cross terms of the
split-operand product on the fp8 pipe; nothing in the product uses it yet.  The lane layout of
v_mfma_scale_f32_16x16x128_f8f6f4 is in no guide: it was probed on the device (tools/probes/fp8_mfma_probe.hip) and is pinned here
against a numpy statement of the arithmetic, on the GPU (developer library) and on the kernel-logic simulator (whose emulation of the
instruction is thereby checked against the same statement)."""
import ctypes

import numpy as np
import pytest
import torch

from visualbert_amd import _lib

pytestmark = pytest.mark.gpu


def e4m3_values():
    """value of each of the 256 OCP e4m3 encodings (NaN for 0x7f / 0xff)"""
    v = np.zeros(256, dtype=np.float64)
    for b in range(256):
        s, e, m = b >> 7, (b >> 3) & 15, b & 7
        if e == 15 and m == 7:
            x = np.nan
        elif e == 0:
            x = m * 2.0 ** -9
        else:
            x = (1 + m / 8.0) * 2.0 ** (e - 7)
        v[b] = -x if s else x
    return v


class _Probe:
    def __init__(self, dev):
        self.dev = dev
        self.ctx = _lib.dev_library() if dev.type == "cuda" else None      # the simulator build carries the probe entries itself

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()
        L = _lib.lib()
        for name in ("vb_mma_f8_probe", "vb_cvt_fp8_probe"):
            fn = getattr(L, name)
            fn.restype, fn.argtypes = _lib.DEV_SIGNATURES[name]
        return L

    def __exit__(self, *exc):
        if self.ctx is not None:
            return self.ctx.__exit__(*exc)
        return False


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_scaled_fp8_mfma_matches_the_documented_layout(dev, seed):
    rng = np.random.default_rng(seed)
    table = e4m3_values()
    finite = np.array([b for b in range(256) if np.isfinite(table[b])], dtype=np.uint8)
    A = rng.choice(finite, size=(16, 128)).astype(np.uint8)
    B = rng.choice(finite, size=(16, 128)).astype(np.uint8)
    if seed == 2:                                            # one-hot rows: a permuted K index or a misplaced scale cannot cancel out
        A[:] = 0
        B[:] = 0x38                                          # 1.0
        for i in range(16):
            A[i, (37 * i + 5) % 128] = 0x38
    sa = rng.integers(120, 135, size=(16, 4)).astype(np.uint8)
    sb = rng.integers(120, 135, size=(16, 4)).astype(np.uint8)
    a = table[A] * np.repeat(2.0 ** (sa.astype(np.float64) - 127), 32, axis=1)
    b = table[B] * np.repeat(2.0 ** (sb.astype(np.float64) - 127), 32, axis=1)
    ref = a @ b.T
    t = lambda x: torch.from_numpy(x).to(dev)
    tA, tB, tsa, tsb = t(A), t(B), t(sa), t(sb)
    D = torch.full((16, 16), float("nan"), dtype=torch.float32, device=dev)
    with _Probe(dev) as L:
        _lib.check(L.vb_mma_f8_probe(_lib.ptr(tA), _lib.ptr(tB), _lib.ptr(tsa), _lib.ptr(tsb), _lib.ptr(D), _lib.stream_ptr()), "vb_mma_f8_probe")
    if dev.type == "cuda":
        torch.cuda.synchronize()
    got = D.cpu().numpy().astype(np.float64)
    # The instruction does not accumulate its 128 products as an exact fp32 chain: with operands spread over e4m3's whole range
    # (2^-9 .. 448, times scales of 2^-7 .. 2^7) MI355X is within 1.1e-4 of the sum of |a||b| (products are aligned to the largest
    # one and truncated); the one-hot case (seed 2) is exact and is what pins the K order and the scale placement.  For the cross
    # terms this primitive is for (2^-9 of the main term) that is 2e-7 of the result.
    scale = np.abs(a) @ np.abs(b).T
    err = float(np.max(np.abs(got - ref) / (scale + 1e-30)))
    assert err <= (0.0 if seed == 2 else 3e-4), err


def test_fp8_conversion_rounds_to_nearest_even(dev):
    table = e4m3_values()
    grid = np.sort(np.unique(table[np.isfinite(table)]))
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(-448, 448, 2000), rng.normal(0, 1, 2000), rng.normal(0, 0.01, 2000), grid,
                        (grid[1:] + grid[:-1]) / 2, [0.0, 448.0, -448.0, 2.0 ** -9, 2.0 ** -10]]).astype(np.float32)
    x = x[: len(x) // 4 * 4]
    tx = torch.from_numpy(x).to(dev)
    ty = torch.zeros(len(x), dtype=torch.uint8, device=dev)
    with _Probe(dev) as L:
        _lib.check(L.vb_cvt_fp8_probe(_lib.ptr(tx), _lib.ptr(ty), len(x), _lib.stream_ptr()), "vb_cvt_fp8_probe")
    if dev.type == "cuda":
        torch.cuda.synchronize()
    got = table[ty.cpu().numpy()]
    # nearest grid point, ties to the even mantissa
    xd = x.astype(np.float64)
    idx = np.searchsorted(grid, xd)
    lo, hi = grid[np.clip(idx - 1, 0, len(grid) - 1)], grid[np.clip(idx, 0, len(grid) - 1)]
    dlo, dhi = np.abs(xd - lo), np.abs(hi - xd)
    code = {float(v): b for b, v in enumerate(table) if np.isfinite(v)}
    even = lambda v: (code[float(v)] & 1) == 0
    want = np.where(dlo < dhi, lo, np.where(dhi < dlo, hi, [l if even(l) else h for l, h in zip(lo, hi)]))
    assert np.array_equal(np.abs(got), np.abs(want)), np.flatnonzero(np.abs(got) != np.abs(want))[:10]
    nz = want != 0
    assert np.array_equal(np.sign(got[nz]), np.sign(want[nz]))
