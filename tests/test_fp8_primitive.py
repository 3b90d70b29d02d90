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


def _bind(L):
    for name in ("vb_split_f8", "vb_gemm_x3f8"):
        fn = getattr(L, name)
        fn.restype, fn.argtypes = _lib.DEV_SIGNATURES[name]


def _split_f8(L, dev, x, K):
    rows, cols = x.shape
    img = torch.zeros(rows, 4 * K, dtype=torch.uint8, device=dev)
    s_hi = torch.zeros((rows + 63) // 64 * 64, dtype=torch.uint8, device=dev)
    s_lo = torch.zeros((rows + 63) // 64 * 64, dtype=torch.uint8, device=dev)
    _lib.check(L.vb_split_f8(_lib.ptr(x), x.stride(0), _lib.ptr(img), 2 * K, rows, cols, _lib.ptr(s_hi), _lib.ptr(s_lo), _lib.stream_ptr()), "vb_split_f8")
    return img, s_hi, s_lo


def _decode(img, s_hi, s_lo, K):
    """fp64 values of the three planes of an image: hi (bf16), hi8 and lo8 (e4m3 x 2^(scale - 127))"""
    table = e4m3_values()
    raw = img.cpu().numpy()
    r = np.arange(raw.shape[0])
    perm = (r & ~63) | ((r & 15) << 2) | ((r >> 4) & 3)      # where vb_split_f8 puts row r's scale
    s_hi, s_lo = s_hi.cpu()[perm], s_lo.cpu()[perm]
    hi = torch.from_numpy(raw[:, :2 * K].copy()).view(torch.bfloat16).float().numpy().astype(np.float64)
    h8 = table[raw[:, 2 * K:3 * K]] * 2.0 ** (s_hi.cpu().numpy().astype(np.float64)[:, None] - 127)
    l8 = table[raw[:, 3 * K:4 * K]] * 2.0 ** (s_lo.cpu().numpy().astype(np.float64)[:, None] - 127)
    return hi, h8, l8


def test_split_f8_planes(dev):
    g = torch.Generator().manual_seed(3)
    rows, cols, K = 37, 200, 256
    x = (torch.randn(rows, cols, generator=g) * torch.exp(torch.randn(rows, 1, generator=g))).to(dev)
    x[5] = 0.0                                                # an all-zero row
    with _Probe(dev) as L:
        _bind(L)
        img, s_hi, s_lo = _split_f8(L, dev, x, K)
    hi, h8, l8 = _decode(img, s_hi, s_lo, K)
    xr = x.cpu().double().numpy()
    want_hi = x.cpu().to(torch.bfloat16).double().numpy()
    assert np.array_equal(hi[:, :cols], want_hi) and not hi[:, cols:].any()
    lo = xr - want_hi
    # every fp8 plane: within half a step of the e4m3 grid of its row's scale (4 significant bits while the element is within 2^-14 of
    # the row's largest; the row's largest itself lands in [224, 448])
    for plane, ref in ((h8, want_hi), (l8, lo)):
        amax = np.abs(ref).max(1, keepdims=True)
        assert np.all(np.abs(plane[:, :cols] - ref) <= np.maximum(np.abs(ref) * 2.0 ** -4, amax * 2.0 ** -17) * 1.0001 + 1e-300)
        assert not plane[:, cols:].any()
    r = np.arange(rows)
    sc = 2.0 ** (127 - s_hi.cpu().numpy()[(r & ~63) | ((r & 15) << 2) | ((r >> 4) & 3)].astype(np.float64))
    nz = np.abs(want_hi).max(1) > 0
    top = np.abs(want_hi).max(1)[nz] * sc[nz]
    assert np.all((top > 223.9) & (top <= 448.0)), (top.min(), top.max())


@pytest.mark.parametrize("shape", [(256, 256, 256), (512, 256, 128), (256, 512, 384)])
def test_gemm_x3f8_prototype(dev, shape):
    """C = hi.hi (bf16 pipe) + lo8.hi8 + hi8.lo8 (fp8 pipe) of the two images, checked (a) against exactly that sum formed in fp64 from
    the images' own bytes -- the kernel's index logic -- and (b) against the fp64 product of the fp32 operands -- what the mode is for"""
    M, N, K = shape
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) * 0.05).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    C = torch.full((M, N), float("nan"), device=dev)
    with _Probe(dev) as L:
        _bind(L)
        ia, ah, al = _split_f8(L, dev, x, K)
        ib, bh, bl = _split_f8(L, dev, w, K)
        _lib.check(L.vb_gemm_x3f8(_lib.ptr(ia), 2 * K, _lib.ptr(ib), 2 * K, _lib.ptr(C), N, M, N, K, _lib.ptr(bias), _lib.ptr(ah), _lib.ptr(al),
                                  _lib.ptr(bh), _lib.ptr(bl), _lib.stream_ptr()), "vb_gemm_x3f8")
    if dev.type == "cuda":
        torch.cuda.synchronize()
    got = C.cpu().double().numpy()
    xh, x8, xl8 = _decode(ia, ah, al, K)
    wh, w8, wl8 = _decode(ib, bh, bl, K)
    b = bias.cpu().double().numpy()
    want = xh @ wh.T + xl8 @ w8.T + x8 @ wl8.T + b
    mag = np.abs(xh) @ np.abs(wh).T + 1.0
    err_logic = float(np.max(np.abs(got - want) / mag))
    exact = x.cpu().double().numpy() @ w.cpu().double().numpy().T + b
    err_exact = float(np.max(np.abs(got - exact) / mag))
    print("x3f8 %s: vs its own planes %.2e, vs the fp64 product %.2e (of sum |x||w|)" % (shape, err_logic, err_exact))
    assert err_logic <= 2e-6, err_logic                       # fp32 accumulation + the fp8 instruction's truncation of 2^-9-sized terms
    assert err_exact <= 3e-4, err_exact                       # the cross terms carry 4 significant bits: ~2^-13 of the product
