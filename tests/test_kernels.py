"""Per-kernel parity: every C-ABI entry point against plain torch fp32 maths on the same inputs.
fp32 kernels must agree to fp32 round-off; bf16 kernels are compared with the same maths on
bf16-rounded inputs at a bf16-sized tolerance (stated per test)."""
import ctypes
import math

import pytest
import torch

from visualbert_amd import _lib

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def tol(dt, f32, bf):
    return f32 if dt == torch.float32 else bf


def gemm(dev, dt, A, B, M, N, K, al, bl, out_f32=False, bias=None, act=0, addend=None, aux_in=None, aux_out=None,
         acc=None, alpha=1.0, alpha_dev=None, colsum=None):
    L = _lib.lib()
    C = acc if acc is not None else torch.full((M, N), 7.0, dtype=torch.float32 if out_f32 else dt, device=dev)
    aux = aux_in if aux_in is not None else aux_out
    rc = L.vb_gemm(_lib.dtype_code(dt), _lib.VB_F32 if out_f32 else _lib.dtype_code(dt), al, bl,
                   _lib.ptr(A), A.stride(0), _lib.ptr(B), B.stride(0), _lib.ptr(C), C.stride(0), M, N, K,
                   alpha, _lib.ptr(alpha_dev), _lib.ptr(bias), _lib.ptr(addend),
                   addend.stride(0) if addend is not None else 0, act, _lib.ptr(aux_in), _lib.ptr(aux_out),
                   aux.stride(0) if aux is not None else 0, 1 if acc is not None else 0, _lib.ptr(colsum),
                   _lib.stream_ptr())
    _lib.check(rc, "vb_gemm")
    return C


def variant_lib(dev, variant):
    """the library that has GEMM kernel `variant`: the product for the kernels its dispatcher can choose, the developer
    library (same objects + experiment arms + the vendor yardstick) for _lib.DEV_NT_KERNELS; the simulator has no dev build."""
    import contextlib
    if variant not in _lib.DEV_NT_KERNELS:
        return contextlib.nullcontext()
    if dev.type != "cuda":
        pytest.skip("experiment arms exist in libvisualbert_hip_dev.so only (no simulator build of it)")
    return _lib.dev_library()


def padded(rows, cols, dt, dev, g):
    """[rows, cols] view of a buffer whose leading dimension is a multiple of 8 (ABI requirement)."""
    ld = (cols + 7) // 8 * 8
    buf = torch.zeros(rows, ld, dtype=dt, device=dev)
    buf[:, :cols] = torch.randn(rows, cols, generator=g).to(dt).to(dev)
    return buf[:, :cols]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(200, 136, 96), (130, 30, 64), (64, 257, 40), (150, 140, 192)])  # last: LDS-direct, 3 K tiles
def test_gemm_forward_layout(dev, dt, shape):
    M, N, K = shape
    g = torch.Generator().manual_seed(1)
    A = padded(M, K, dt, dev, g)
    B = padded(N, K, dt, dev, g)
    bias = torch.randn(N, generator=g).to(dev)
    C = gemm(dev, dt, A, B, M, N, K, 0, 0, bias=bias)
    ref = A.float() @ B.float().t() + bias
    err = (C.float() - ref).abs().max().item()
    assert err <= tol(dt, 2e-4, 0.02 * ref.abs().max().item()), err
    # fp32 output + GELU + pre-activation + addend
    add = padded(M, N, dt, dev, g)
    aux = torch.zeros(M, (N + 7) // 8 * 8, dtype=dt, device=dev)[:, :N]
    C = gemm(dev, dt, A, B, M, N, K, 0, 0, out_f32=True, bias=bias, act=_lib.VB_ACT_GELU, addend=add, aux_out=aux)
    ref2 = torch.nn.functional.gelu(ref) + add.float()
    assert (C - ref2).abs().max().item() <= 2e-4
    assert (aux.float() - ref).abs().max().item() <= tol(dt, 2e-4, 0.02 * ref.abs().max().item())
    # tanh
    C = gemm(dev, dt, A, B, M, N, K, 0, 0, out_f32=True, bias=bias, act=_lib.VB_ACT_TANH, alpha=0.1)
    assert (C - torch.tanh(0.1 * (ref - bias) + bias)).abs().max().item() <= 2e-5


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(200, 136, 72), (130, 30, 40), (70, 300, 136), (1100, 136, 72)])  # last: split-K wgrad
def test_gemm_dgrad_wgrad_layouts(dev, dt, shape):
    M, N, K2 = shape          # dY [M,N], W [N,K2], X [M,K2]
    g = torch.Generator().manual_seed(2)
    dY = padded(M, N, dt, dev, g)
    W = padded(N, K2, dt, dev, g)
    X = padded(M, K2, dt, dev, g)
    pre = padded(M, K2, dt, dev, g)
    # dgrad: dX = dY W  (B operand K-strided), fused GELU' and residual-gradient addend
    add = padded(M, K2, dt, dev, g)
    cs = torch.ones(K2, device=dev)
    C = gemm(dev, dt, dY, W, M, K2, N, 0, 1, out_f32=True, act=_lib.VB_ACT_GELU_GRAD, aux_in=pre, addend=add, colsum=cs)
    assert (cs - (1.0 + C.sum(0))).abs().max().item() <= 2e-3 * max(1.0, C.sum(0).abs().max().item())   # fused bias gradient
    x = pre.float()
    gelu_grad = 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
    ref = (dY.float() @ W.float()) * gelu_grad + add.float()
    assert (C - ref).abs().max().item() <= 3e-4 * max(1.0, ref.abs().max().item())
    # wgrad: dW += alpha_dev * dY^T X  (both operands K-strided, fp32 accumulate in place)
    acc = torch.ones(N, K2, device=dev)
    sc = torch.tensor([0.5], device=dev)
    C = gemm(dev, dt, dY, X, N, K2, M, 1, 1, out_f32=True, acc=acc, alpha_dev=sc)
    ref = 0.5 * (dY.float().t() @ X.float()) + 1.0
    assert (C - ref).abs().max().item() <= 3e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("shape,addend", [((100, 128, 1536), False), ((100, 128, 1536), True), ((1312, 768, 3072), False),
                                          ((1312, 768, 2304), True), ((650, 768, 3072), True)])
def test_gemm_split_k_across_compute_units(dev, shape, addend):
    """Small long-K problems (per-GPU batch 8: M = 1312, N = 768, K = 3072 / 2304) leave half of the chip idle: the reduction is cut into
    slices on different compute units, partial tiles go through the stream's scratch (vb_stream_set_scratch), the last slice to arrive
    sums them in slice order and runs the epilogue.  Against fp32 matmul, bias and "+ addend" epilogues; the SAME bits on every repeat
    (the summation order does not depend on which slice arrives last) and the same bits as the unsplit kernel up to fp32 reassociation."""
    M, N, K = shape
    if dev.type != "cuda" and M > 128:
        pytest.skip("simulator: the small case covers the index logic (4 compute units)")
    g = torch.Generator().manual_seed(M + K)
    dt = torch.bfloat16
    A = (0.5 * torch.randn(M, K, generator=g)).to(dt).to(dev)
    W = (0.05 * torch.randn(N, K, generator=g)).to(dt).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    add = torch.randn(M, N, generator=g).to(dt).to(dev) if addend else None
    ref = A.float() @ W.float().t() + bias + (add.float() if addend else 0.0)
    outs = []
    L = _lib.lib()
    for rep in range(3):
        L.vb_stream_profile(_lib.stream_ptr(), 1)
        C = gemm(dev, dt, A, W, M, N, K, 0, 0, bias=bias, addend=add)
        if dev.type == "cuda":
            torch.cuda.synchronize()
            n = 8
            ms, fl, ky = (ctypes.c_double * n)(), (ctypes.c_double * n)(), (ctypes.c_int * n)()
            cnt = L.vb_stream_profile_read(_lib.stream_ptr(), ms, fl, ky, n)
            assert cnt == 1 and (ky[0] & 1024), [ky[i] for i in range(max(cnt, 0))]        # the split-K instantiation ran
        L.vb_stream_profile(_lib.stream_ptr(), 0)
        assert (C.float() - ref).abs().max().item() <= 1e-2 * max(1.0, ref.abs().max().item())
        outs.append(C.clone())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    with _lib.stream_opts(nt_kernel=22):                      # a pinned kernel is never split
        C22 = gemm(dev, dt, A, W, M, N, K, 0, 0, bias=bias, addend=add)
    assert (C22.float() - outs[0].float()).abs().max().item() <= 2e-2 * max(1.0, ref.abs().max().item())
    # the hand-off under reuse: the slabs of launch i hold launch i - 1's partial tiles when launch i starts, and the reader may sit on
    # another XCD than the writers -- alternate two inputs for many launches; a stale slab line (or a ticket that overtook its slab)
    # would reproduce the OTHER input's partial sums somewhere.  Each result must be the bits its input gave the first time.
    if dev.type == "cuda":
        A2 = (0.5 * torch.randn(M, K, generator=g)).to(dt).to(dev)
        want = [outs[0], gemm(dev, dt, A2, W, M, N, K, 0, 0, bias=bias, addend=add).clone()]
        big = torch.randn(64 << 20, device=dev)               # 256 MB stream between launches: uneven load, L2 contents churned
        bad = 0
        for i in range(120):
            if i % 3 == 0:
                big.mul_(1.0001)
            Ci = gemm(dev, dt, A2 if i & 1 else A, W, M, N, K, 0, 0, bias=bias, addend=add)
            bad += int(not torch.equal(Ci, want[i & 1]))
        assert bad == 0, "%d of 120 split-K launches differ from their input's first result" % bad


def test_split_k_scratch_is_per_stream(dev):
    """the partial tiles and arrival counters of the split-K GEMMs live in scratch that belongs to a STREAM (vb_stream_set_scratch;
    _lib.stream_ptr registers one buffer per stream on first use): two streams running split-K GEMMs at the same time must not see each
    other's slabs or tickets.  Two different problems, 40 launches each, interleaved on two streams; every result bit-equal to the
    problem's result computed alone."""
    if dev.type != "cuda":
        pytest.skip("needs two HIP streams")
    g = torch.Generator().manual_seed(99)
    dt = torch.bfloat16
    M, N, K = 1312, 768, 3072
    probs = []
    for i in range(2):
        A = (0.5 * torch.randn(M, K, generator=g)).to(dt).to(dev)
        W = (0.05 * torch.randn(N, K, generator=g)).to(dt).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        probs.append((A, W, bias, gemm(dev, dt, A, W, M, N, K, 0, 0, bias=bias).clone()))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    outs = [[], []]
    for rep in range(40):
        for i, st in enumerate(streams):
            with torch.cuda.stream(st):
                A, W, bias, _ = probs[i]
                outs[i].append(gemm(dev, dt, A, W, M, N, K, 0, 0, bias=bias))
    torch.cuda.synchronize()
    for i in range(2):
        bad = sum(int(not torch.equal(c, probs[i][3])) for c in outs[i])
        assert bad == 0, "stream %d: %d of 40 results differ" % (i, bad)
    assert len({k for k in _lib._scratch if k[2] in (streams[0].cuda_stream, streams[1].cuda_stream)}) == 2   # one buffer per stream


def _wgrad_grouped(dev, dys, xs, dws, tokens, alpha=1.0, alpha_dev=None):
    L = _lib.lib()
    n = len(dys)
    PA = ctypes.c_void_p * n
    I64 = ctypes.c_int64 * n
    I32 = ctypes.c_int * n
    rc = L.vb_wgrad_grouped(_lib.VB_BF16, n, PA(*[_lib.ptr(t) for t in dys]), I64(*[t.stride(0) for t in dys]),
                            PA(*[_lib.ptr(t) for t in xs]), I64(*[t.stride(0) for t in xs]),
                            PA(*[_lib.ptr(t) for t in dws]), I64(*[t.stride(0) for t in dws]),
                            I32(*[t.size(1) for t in dys]), I32(*[t.size(1) for t in xs]), tokens, alpha,
                            _lib.ptr(alpha_dev), _lib.stream_ptr())
    _lib.check(rc, "vb_wgrad_grouped")


@pytest.mark.parametrize("tokens", [64, 192, 704, 100, 749])    # 100, 749: ragged token counts (B x S not a multiple of 64): whole
@pytest.mark.parametrize("wgs", [0, 1, 3])                       # K tiles on the grouped kernel, the last 36 / 45 tokens on the grouped tail kernel (one launch for all problems)
def test_wgrad_grouped_transposing_reads(dev, tokens, wgs):
    """dW += dY^T X for a group of Linears in one persistent launch: operands copied as stored ([token][feature]) and
    gathered into MFMA fragments by ds_read_b64_tr_b16 (in the simulator: the measured lane map).  1, 3 and 11 K tiles
    of 64 tokens, ragged 256x256 output tiles, token slices added atomically, one / three workgroups walking all
    items; also the same shapes through vb_gemm's K-strided x K-strided entry (the per-op path)."""
    L = _lib.lib()
    g = torch.Generator().manual_seed(tokens + wgs)
    dt = torch.bfloat16
    shapes = [(264, 520), (768, 256), (8, 72), (282, 45)]       # (out, in); the last: neither a multiple of 8
    dys = [padded(tokens, o, dt, dev, g) for o, _ in shapes]
    xs = [padded(tokens, i, dt, dev, g) for _, i in shapes]
    with _lib.stream_opts(persistent_workgroups=wgs):
        dws = [torch.full((o, i), 1.5, device=dev) for o, i in shapes]
        sc = torch.tensor([0.5], device=dev)
        _wgrad_grouped(dev, dys, xs, dws, tokens, alpha=2.0, alpha_dev=sc)
        for dy, x, dw in zip(dys, xs, dws):
            ref = 1.5 + dy.float().t() @ x.float()
            assert (dw - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())
        # per-op entry: vb_gemm(K-strided, K-strided, fp32 accumulate)
        acc = torch.full(shapes[0], -1.0, device=dev)
        C = gemm(dev, dt, dys[0], xs[0], shapes[0][0], shapes[0][1], tokens, 1, 1, out_f32=True, acc=acc)
        ref = -1.0 + dys[0].float().t() @ xs[0].float()
        assert (C - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("tokens", [24, 64, 100, 448, 1312, 3072])   # 24: only a ragged tile; 1312 = per-GPU batch 8 x 164; 3072 = 48 K tiles (the crossover)
@pytest.mark.parametrize("wgs", [0, 8])                            # 0: tiles <= compute units (four-stage ring); 8: more tiles than units (two stages, two workgroups per CU)
def test_wgrad_small_token_kernel(dev, tokens, wgs):
    """gemm_tn_small_kernel: dW += alpha dY^T X for a group of Linears at SMALL token counts -- one workgroup per 128x128 tile of dW walks the
    whole token range (whole 64-token K tiles by LDS-direct copies, the ragged rest through registers with zero fill) and adds the tile
    with plain read-modify-writes.  Ragged output rows (Mo not a multiple of 128 or of 8), several problems, both ring depths, alpha from
    the device; twice into the same dW (no stale state between launches)."""
    if dev.type != "cuda" and tokens > 448:
        pytest.skip("simulator: the small cases cover the index logic")
    g = torch.Generator().manual_seed(tokens + wgs)
    dt = torch.bfloat16
    shapes = [(264, 520), (768, 256), (8, 72), (77, 136), (128, 128)]          # (out, in): in % 8 == 0 (the kernel's vector rows)
    dys = [padded(tokens, o, dt, dev, g) for o, _ in shapes]
    xs = [padded(tokens, i, dt, dev, g) for _, i in shapes]
    with _lib.stream_opts(persistent_workgroups=wgs):
        _lib.lib().vb_stream_profile(_lib.stream_ptr(), 1)
        dws = [torch.full((o, i), 1.5, device=dev) for o, i in shapes]
        sc = torch.tensor([0.5], device=dev)
        _wgrad_grouped(dev, dys, xs, dws, tokens, alpha=2.0, alpha_dev=sc)
        _wgrad_grouped(dev, dys, xs, dws, tokens, alpha=-1.0)
        if dev.type == "cuda":
            torch.cuda.synchronize()
        n = 64
        ms, fl, ky = (ctypes.c_double * n)(), (ctypes.c_double * n)(), (ctypes.c_int * n)()
        cnt = _lib.lib().vb_stream_profile_read(_lib.stream_ptr(), ms, fl, ky, n)
        _lib.lib().vb_stream_profile(_lib.stream_ptr(), 0)
        if dev.type == "cuda":                      # (the simulator build records no launches)
            assert cnt == 2 and all((ky[i] & 255) == 39 for i in range(cnt)), [ky[i] for i in range(max(cnt, 0))]   # ONE launch per call, the small-token kernel
        for dy, x, dw in zip(dys, xs, dws):
            ref = 1.5 + 0.0 * (dy.float().t() @ x.float())
            assert (dw - ref).abs().max().item() <= 4e-4 * max(1.0, (dy.float().t() @ x.float()).abs().max().item())
        dws = [torch.full((o, i), 1.5, device=dev) for o, i in shapes]
        _wgrad_grouped(dev, dys, xs, dws, tokens, alpha=2.0, alpha_dev=sc)
        for dy, x, dw in zip(dys, xs, dws):
            ref = 1.5 + dy.float().t() @ x.float()
            assert (dw - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("tokens", [40, 200, 1312])
def test_wgrad_small_token_kernel_256x128_tiles(dev, tokens):
    """gemm_tn_small256_kernel: the eight-wave 256x128-tile form of the small-token weight gradients, taken when the 128x128 tiles
    outnumber the compute units but the 256x128 tiles fit in one round -- an encoder layer's four Linears on the real chip (216
    tiles), one ragged 300 x 200 problem on the simulator's four units.  Against fp32 matmul, accumulating twice."""
    if dev.type == "cuda":
        shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
    else:
        if tokens > 200:
            pytest.skip("simulator: the small cases cover the index logic")
        shapes = [(300, 200)]
    g = torch.Generator().manual_seed(tokens)
    dt = torch.bfloat16
    dys = [padded(tokens, o, dt, dev, g) for o, _ in shapes]          # (leading dimensions rounded up to 8: the ABI's requirement)
    xs = [padded(tokens, i, dt, dev, g) for _, i in shapes]
    dws = [torch.full((o, i), 0.25, device=dev) for o, i in shapes]
    sc = torch.tensor([0.5], device=dev)
    _wgrad_grouped(dev, dys, xs, dws, tokens, alpha=2.0, alpha_dev=sc)
    _wgrad_grouped(dev, dys, xs, dws, tokens, alpha=1.0)
    for dy, x, dw in zip(dys, xs, dws):
        ref = 0.25 + 2.0 * (dy.float().t() @ x.float())
        assert (dw - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())


def test_wgrad_grouped_problems_sharing_one_dw(dev):
    """two problems of one call that accumulate into the SAME dW (the ABI does not forbid it): the grouped kernel adds with atomics;
    the ragged rows' tail kernel uses plain read-modify-writes and must therefore run such problems one after the other."""
    g = torch.Generator().manual_seed(77)
    dt = torch.bfloat16
    tokens, o, i = 100, 72, 136                                  # one K tile + 36 ragged rows
    dys = [padded(tokens, o, dt, dev, g) for _ in range(2)]
    xs = [padded(tokens, i, dt, dev, g) for _ in range(2)]
    dw = torch.full((o, i), 0.5, device=dev)
    _wgrad_grouped(dev, dys, xs, [dw, dw], tokens)
    ref = 0.5 + dys[0].float().t() @ xs[0].float() + dys[1].float().t() @ xs[1].float()
    assert (dw - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())


def test_wgrad_grouped_helper_layout_on_the_full_chip(dev):
    """An encoder layer's four weight gradients are 108 tiles: two token slices each leave 40 of 256 compute units idle, so
    the launch switches to the helper layout (slices over the first KT - rem K tiles, the idle CUs take the last `rem` of
    tiles h, h + 40, ...).  BERT-base shapes at 20,992 tokens against fp32 matmuls, twice (atomics: no stale state)."""
    if dev.type != "cuda":
        pytest.skip("needs the real chip's workgroup count (the simulator runs 4)")
    tokens = 20992
    g = torch.Generator().manual_seed(5)
    dt = torch.bfloat16
    shapes = [(768, 3072), (3072, 768), (768, 768), (2304, 768)]
    dys = [(0.3 * torch.randn(tokens, o, generator=g)).to(dt).to(dev) for o, _ in shapes]
    xs = [(0.3 * torch.randn(tokens, i, generator=g)).to(dt).to(dev) for _, i in shapes]
    for rep in range(2):
        dws = [torch.full((o, i), 0.25, device=dev) for o, i in shapes]
        _wgrad_grouped(dev, dys, xs, dws, tokens)
        for dy, x, dw in zip(dys, xs, dws):
            ref = 0.25 + dy.float().t() @ x.float()
            assert (dw - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("V", [1000, 33000])        # register-cached single pass (V <= 32768) / rolled two-pass rows
def test_cross_entropy_dense_and_compact_rows(dev, dt, V):
    """CrossEntropyLoss(ignore_index=-1) + gradient, dense rows (vb_ce_fwd_bwd) and the compact form the MLM head uses
    (vb_ce_fwd_bwd_rows: gradient of the labelled rows only, zero padding rows) against torch."""
    L = _lib.lib()
    M = 37
    g = torch.Generator().manual_seed(V)
    ld = (V + 63) // 64 * 64
    logits = torch.zeros(M, ld, device=dev)
    logits[:, :V] = (torch.randn(M, V, generator=g) * 3).to(dev)
    lab = torch.randint(0, V, (M,), generator=g)
    lab[torch.rand(M, generator=g) < 0.6] = -1
    lab[3] = V - 1                                   # last valid column
    lab = lab.to(dev)
    ref_in = logits[:, :V].detach().clone().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(ref_in, lab, ignore_index=-1)
    ref.backward()
    acc = torch.empty(66, device=dev); loss = torch.empty(1, device=dev)
    dl = torch.full((M, ld), 7.0, dtype=dt, device=dev)
    _lib.check(L.vb_ce_fwd_bwd(_lib.dtype_code(dt), _lib.ptr(logits), ld, _lib.ptr(lab), -1, _lib.ptr(acc), _lib.ptr(loss),
                               _lib.ptr(dl), ld, M, V, _lib.stream_ptr()), "vb_ce_fwd_bwd")
    assert abs(loss.item() - ref.item()) <= 2e-5 * max(1.0, abs(ref.item()))
    t = tol(dt, 2e-6, 2e-3)
    assert (dl[:, :V].float() - ref_in.grad).abs().max().item() <= t
    assert (dl[:, V:].float() == 0).all()                                   # pad columns are written as zeros
    rows = torch.nonzero(lab != -1).reshape(-1)
    n = rows.numel(); n_pad = (n + 63) // 64 * 64
    dlc = torch.full((n_pad, ld), 7.0, dtype=dt, device=dev)
    _lib.check(L.vb_ce_fwd_bwd_rows(_lib.dtype_code(dt), _lib.ptr(logits), ld, _lib.ptr(lab), -1, _lib.ptr(rows), n, n_pad,
                                    _lib.ptr(acc), _lib.ptr(loss), _lib.ptr(dlc), ld, M, V, _lib.stream_ptr()),
               "vb_ce_fwd_bwd_rows")
    assert abs(loss.item() - ref.item()) <= 2e-5 * max(1.0, abs(ref.item()))
    assert (dlc[:n, :V].float() - ref_in.grad[rows]).abs().max().item() <= t
    assert (dlc[n:].float() == 0).all() and (dlc[:, V:].float() == 0).all()


def ln_fwd(dev, dt, x, resid, gamma, beta, p_in=0.0, p_out=0.0, seed=5, want_z=True):
    M, H = x.shape
    L = _lib.lib()
    y = torch.empty_like(x)
    z = torch.empty_like(x) if want_z else None
    mean = torch.empty(M, device=dev)
    rstd = torch.empty(M, device=dev)
    rc = L.vb_ln_fwd(_lib.dtype_code(dt), _lib.ptr(x), _lib.ptr(resid), _lib.ptr(z), _lib.ptr(y), _lib.ptr(mean),
                     _lib.ptr(rstd), _lib.ptr(gamma), _lib.ptr(beta), M, H, 1e-12, p_in, 11, p_out, 12, seed,
                     _lib.stream_ptr())
    _lib.check(rc, "vb_ln_fwd")
    return y, z, mean, rstd


def ln_bwd(dev, dt, dy, z, mean, rstd, gamma, p_in=0.0, p_out=0.0, seed=5, two_stage=True):
    M, H = dy.shape
    L = _lib.lib()
    dz = torch.empty_like(dy)
    dx = torch.empty_like(dy)
    dg = torch.zeros(H, device=dev)
    db = torch.zeros(H, device=dev)
    dbias = torch.zeros(H, device=dev)
    ws = torch.empty(L.vb_ln_bwd_ws_bytes(M, H) // 4, device=dev) if two_stage else None
    rc = L.vb_ln_bwd(_lib.dtype_code(dt), _lib.ptr(dy), _lib.ptr(z), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(gamma),
                     _lib.ptr(dz), _lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db), _lib.ptr(dbias), M, H,
                     p_in, 11, p_out, 12, seed,
                     _lib.ptr(ws),
                     _lib.stream_ptr())
    _lib.check(rc, "vb_ln_bwd")
    return dz, dx, dg, db, dbias


def ref_ln(z, gamma, beta):
    u = z.mean(-1, keepdim=True)
    s = (z - u).pow(2).mean(-1, keepdim=True)
    return gamma * ((z - u) / torch.sqrt(s + 1e-12)) + beta


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("MH", [(37, 128), (21, 768), (13, 640), (9, 1024), (2100, 768)])   # 640: a partly filled 4-column chunk; last: sliced second-stage reduction
def test_layernorm_residual_fwd_bwd(dev, dt, MH):
    M, H = MH
    g = torch.Generator().manual_seed(3)
    x = torch.randn(M, H, generator=g).to(dt).to(dev)
    r = torch.randn(M, H, generator=g).to(dt).to(dev)
    gamma = (1 + 0.1 * torch.randn(H, generator=g)).to(dev)
    beta = (0.1 * torch.randn(H, generator=g)).to(dev)
    y, z, mean, rstd = ln_fwd(dev, dt, x, r, gamma, beta)
    zr = (x.float() + r.float()).requires_grad_(True)
    yr = ref_ln(zr, gamma, beta)
    assert (y.float() - yr).abs().max().item() <= tol(dt, 2e-5, 0.04)
    assert (z.float() - zr).abs().max().item() <= tol(dt, 1e-6, 0.04)
    dy = torch.randn(M, H, generator=g).to(dt).to(dev)
    # backward consumes the stored (T-rounded) z: the reference does the same
    zs = z.float().detach().requires_grad_(True)
    gp = gamma.clone().requires_grad_(True)
    bp = beta.clone().requires_grad_(True)
    ref_ln(zs, gp, bp).backward(dy.float())
    dz, dx, dg, db, dbias = ln_bwd(dev, dt, dy, z, mean, rstd, gamma)
    _, _, dg2, db2, dbias2 = ln_bwd(dev, dt, dy, z, mean, rstd, gamma, two_stage=False)   # atomics path
    for u, v in ((dg, dg2), (db, db2), (dbias, dbias2)):
        assert (u - v).abs().max().item() <= 1e-4 * max(1.0, v.abs().max().item())
    t = tol(dt, 5e-5, 0.05)
    assert (dz.float() - zs.grad).abs().max().item() <= t * max(1.0, zs.grad.abs().max().item())
    assert torch.equal(dx, dz)
    assert (dg - gp.grad).abs().max().item() <= tol(dt, 2e-4, 0.02) * max(1.0, gp.grad.abs().max().item())
    assert (db - bp.grad).abs().max().item() <= 2e-4 * max(1.0, bp.grad.abs().max().item())
    # dbias is summed from the fp32 dz (before rounding to T)
    assert (dbias - zs.grad.sum(0)).abs().max().item() <= tol(dt, 2e-4, 0.01) * max(1.0, zs.grad.sum(0).abs().max().item())


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("H", [768, 640, 256])        # 768 / 640: the 8 + 4-columns-per-lane backward (640: a partly filled 4-column chunk)
def test_layernorm_dropout_consistency(dev, dt, H):
    """dropout masks are regenerated, not stored: forward and backward must agree on them, the keep
    rate must be 1-p, and different stream ids must give different masks."""
    M, p = 64, 0.1
    g = torch.Generator().manual_seed(4)
    x = (1.0 + torch.rand(M, H, generator=g)).to(dt).to(dev)        # strictly positive: zeros == dropped
    zero = torch.zeros(M, H, dtype=dt, device=dev)
    gamma = torch.ones(H, device=dev)
    beta = torch.zeros(H, device=dev)
    # input dropout: z = drop(x) + 0
    y, z, mean, rstd = ln_fwd(dev, dt, x, zero, gamma, beta, p_in=p)
    keep_in = z.float() != 0
    rate = keep_in.float().mean().item()
    assert abs(rate - (1 - p)) < 0.01, rate
    assert (z.float()[keep_in] - (x.float() / (1 - p))[keep_in]).abs().max().item() <= tol(dt, 1e-5, 0.02)
    dy = torch.ones(M, H, dtype=dt, device=dev)
    dz, dx, *_ = ln_bwd(dev, dt, dy * 0 + torch.randn(M, H, generator=g).to(dt).to(dev), z, mean, rstd, gamma, p_in=p)
    assert torch.equal(dx.float() != 0, keep_in & (dz.float() != 0))
    sel = keep_in & (dz.float() != 0)
    assert (dx.float()[sel] - dz.float()[sel] / (1 - p)).abs().max().item() <= tol(dt, 1e-5, 0.02) * 5
    # output dropout: y = drop(LN(x))
    y2, _, _, _ = ln_fwd(dev, dt, x, None, gamma, beta, p_out=p, want_z=False)
    y0, _, _, _ = ln_fwd(dev, dt, x, None, gamma, beta, want_z=False)
    keep_out = y2.float() != 0
    assert abs(keep_out.float().mean().item() - (1 - p)) < 0.01
    assert (keep_out != keep_in).any()                               # stream ids 11 vs 12 differ
    assert (y2.float()[keep_out] - y0.float()[keep_out] / (1 - p)).abs().max().item() <= tol(dt, 1e-5, 0.05)
    # a different seed gives a different mask
    _, z3, _, _ = ln_fwd(dev, dt, x, zero, gamma, beta, p_in=p, seed=6)
    assert ((z3.float() != 0) != keep_in).any()


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("B", [5, 19])          # 19: the backward kernel slices the batch over gridDim.y
def test_embedding_fwd_bwd(dev, dt, B):
    T, R, H, V, TV, P = 12, 6, 128, 50, 2, 64
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, V, (B, T), generator=g).to(dev)
    tt = torch.randint(0, TV, (B, T), generator=g).to(dev)
    vt = torch.randint(0, TV, (B, R), generator=g).to(dev)
    word = torch.randn(V, H, generator=g).to(dev)
    pos = torch.randn(P, H, generator=g).to(dev)
    typ = torch.randn(TV, H, generator=g).to(dev)
    posv = torch.randn(P, H, generator=g).to(dev)
    typv = torch.randn(TV, H, generator=g).to(dev)
    vp = torch.randn(B * R, H, generator=g).to(dt).to(dev)
    z = torch.empty(B, T + R, H, dtype=dt, device=dev)
    pal = torch.randn(B * R, H, generator=g).to(dev)           # vb_align_pos_fwd's output (image_text_alignment branch)
    L = _lib.lib()
    text = word[ids] + pos[:T].unsqueeze(0) + typ[tt]
    for pos_align in (None, pal):
        rc = L.vb_embed_fwd(_lib.dtype_code(dt), _lib.ptr(ids), _lib.ptr(tt), _lib.ptr(vt), _lib.ptr(vp), _lib.ptr(word),
                            _lib.ptr(pos), _lib.ptr(typ), _lib.ptr(posv), _lib.ptr(typv), _lib.ptr(pos_align), _lib.ptr(z),
                            B, T, R, H, V, TV, P, _lib.stream_ptr())
        _lib.check(rc, "vb_embed_fwd")
        vis = vp.float().view(B, R, H) + posv[0] + typv[vt]
        if pos_align is not None:
            vis = vis + pos_align.view(B, R, H)
        ref = torch.cat((text, vis), 1)
        assert (z.float() - ref).abs().max().item() <= tol(dt, 2e-6, 0.04)
    dz = torch.randn(B, T + R, H, generator=g).to(dt).to(dev)
    dw = torch.ones(V, H, device=dev); dp = torch.ones(P, H, device=dev); dty = torch.ones(TV, H, device=dev)
    dpv = torch.ones(P, H, device=dev); dtv = torch.ones(TV, H, device=dev)
    dvp = torch.empty(B * R, H, dtype=dt, device=dev)
    rc = L.vb_embed_bwd(_lib.dtype_code(dt), _lib.ptr(dz), _lib.ptr(ids), _lib.ptr(tt), _lib.ptr(vt), _lib.ptr(dw),
                        _lib.ptr(dp), _lib.ptr(dty), _lib.ptr(dpv), _lib.ptr(dtv), _lib.ptr(dvp), B, T, R, H, V, TV, P,
                        _lib.stream_ptr())
    _lib.check(rc, "vb_embed_bwd")
    d = dz.float()
    rw = torch.ones(V, H, device=dev).index_put_((ids.reshape(-1),), d[:, :T].reshape(-1, H), accumulate=True)
    rp = torch.ones(P, H, device=dev); rp[:T] += d[:, :T].sum(0)
    rt = torch.ones(TV, H, device=dev).index_put_((tt.reshape(-1),), d[:, :T].reshape(-1, H), accumulate=True)
    rpv = torch.ones(P, H, device=dev); rpv[0] += d[:, T:].sum((0, 1))
    rtv = torch.ones(TV, H, device=dev).index_put_((vt.reshape(-1),), d[:, T:].reshape(-1, H), accumulate=True)
    for a, b in ((dw, rw), (dp, rp), (dty, rt), (dpv, rpv), (dtv, rtv)):
        assert (a - b).abs().max().item() <= 1e-4
    assert torch.equal(dvp.view(B, R, H), dz[:, T:])


# bf16 attention kernels against the fp32 restatement on the SAME bf16 inputs and the kernel's own keep-bits: 1.5 x the largest
# error measured on MI355X over every configuration of test_attention_fwd_bwd (profiles/r03_parity_small.json,
# attention_kernel_bf16: ctx <= 2.4e-3 absolute, dqkv <= 4.1e-3 of max |grad| -- round 2 asserted 3e-2 / 4e-2, wide enough to
# hide a mis-scaled fragment; fp32 kernels: 2.5e-7 / 5.4e-7 measured, asserted at 2e-6 / 3e-6)
ATTN_BF16_CTX, ATTN_BF16_DQKV = 3.6e-3, 6.2e-3
# split mode (VB_BF16X3: fp32 tensors, three bf16 MFMAs per product): same tests, fp32 tensors, the "x3" entry of ATTN_MODES
ATTN_MODES = DTYPES + ["x3"]
ATTN_X3_CTX, ATTN_X3_DQKV = 1.2e-5, 1.6e-5      # 6.1e-6 / 8.5e-6 at worst over the configurations below


def attn_mode(dt):
    """(tensor dtype, dtype code of the C-ABI, tolerance picker) of an ATTN_MODES entry."""
    if dt == "x3":
        return torch.float32, _lib.VB_BF16X3, lambda f32, bf, x3: x3
    return dt, _lib.dtype_code(dt), lambda f32, bf, x3: tol(dt, f32, bf)


def attn_ref(qkv, mask_add, nh, keep=None, p=0.0):
    """plain fp32 restatement of modeling.py:236-256 on packed qkv [B,S,3H]."""
    B, S, H3 = qkv.shape
    H = H3 // 3
    d = H // nh
    q, k, v = [x.view(B, S, nh, d).permute(0, 2, 1, 3) for x in qkv.split(H, dim=-1)]
    s = q @ k.transpose(-1, -2) / math.sqrt(d) + mask_add.view(B, 1, 1, S)
    pr = torch.softmax(s, -1)
    if keep is not None:
        pr = pr * keep / (1 - p)
    ctx = (pr @ v).permute(0, 2, 1, 3).reshape(B, S, H)
    return ctx, torch.logsumexp(s, -1)


def decode_keepbits(bits, B, nh, S):
    """uint64 keep-bits -> bool [B,nh,S(query),S(key)] (layout documented in attention.hip)."""
    nw = bits.numel() // (B * nh * S * 4)
    w = bits.view(B * nh, S, 4, nw).cpu()
    keep = torch.zeros(B * nh, S, S, dtype=torch.bool)
    for key in range(S):
        kf, g, r = key // 16, (key % 16) // 4, key % 4
        word = w[:, :, g, kf // 16]
        keep[:, :, key] = ((word >> ((kf % 16) * 4 + r)) & 1).bool()
    return keep.view(B, nh, S, S)


@pytest.mark.parametrize("dt", ATTN_MODES)
@pytest.mark.parametrize("cfg", [(2, 40, 2, 0.0), (1, 164, 2, 0.0), (2, 23, 3, 0.1), (1, 164, 1, 0.1),
                                 (2, 56, 2, 0.1), (1, 112, 2, 0.1),      # VQA / NLVR2 lengths: exact 4- and 7-fragment forwards
                                 (1, 192, 1, 0.1), (2, 177, 2, 0.0),     # edges of the one-pass bf16 backward (12 key fragments)
                                 (1, 300, 1, 0.1), (1, 416, 1, 0.1),     # 416 = NLVR2 as the reference runs it (2x144 + 128)
                                 (2, 230, 1, 0.1), (1, 512, 2, 0.1)])    # key-tiled kernels: fp32 dQ pass above 192 keys, fp32 forward
                                                                         # above 256, bf16 dQ pass above 416; 512 = max_position_embeddings
def test_attention_fwd_bwd(dev, dt, cfg):
    B, S, nh, p = cfg
    tag = {torch.bfloat16: "bf16", torch.float32: "fp32", "x3": "bf16x3"}[dt]
    dt, code, pick = attn_mode(dt)
    H = nh * 64
    g = torch.Generator().manual_seed(6)
    qkv = (0.7 * torch.randn(B, S, 3 * H, generator=g)).to(dt).to(dev)
    lens = torch.randint(S // 2, S + 1, (B,), generator=g)
    mask = (torch.arange(S).unsqueeze(0) < lens.unsqueeze(1)).float()
    mask[:, S // 3] = 0                                             # a masked slot in the middle (padded text)
    mask_add = ((1 - mask) * -10000.0).to(dev)
    L = _lib.lib()
    ctx = torch.empty(B, S, H, dtype=dt, device=dev)
    lse = torch.empty(B, nh, S, device=dev)
    nwords = L.vb_attn_keepbits_words(S)
    bits = torch.zeros(B * nh * nwords, dtype=torch.int64, device=dev)
    rc = L.vb_attn_fwd(code, _lib.ptr(qkv), _lib.ptr(mask_add), _lib.ptr(ctx), _lib.ptr(lse),
                       _lib.ptr(bits), B, S, nh, 64, p, 77, 3, _lib.stream_ptr())
    _lib.check(rc, "vb_attn_fwd")
    keep = None
    if p > 0:
        keep = decode_keepbits(bits, B, nh, S).to(dev).float()
        valid = keep[:, :, :, :]
        rate = valid.mean().item()
        assert abs(rate - (1 - p)) < 0.02, rate
    qr = qkv.float().detach().requires_grad_(True)
    ctx_r, lse_r = attn_ref(qr, mask_add, nh, keep, p)
    t = pick(2e-6, ATTN_BF16_CTX, ATTN_X3_CTX)
    ctx_err = (ctx.float() - ctx_r).abs().max().item()
    assert ctx_err <= t, ctx_err
    assert (lse - lse_r).abs().max().item() <= pick(2e-5, 2e-3, 4e-5)
    dctx = torch.randn(B, S, H, generator=g).to(dt).to(dev)
    ctx_r.backward(dctx.float())
    dqkv = torch.full((B, S, 3 * H), float("nan"), dtype=dt, device=dev)
    ws = torch.empty(L.vb_attn_bwd_ws_floats(B, S, nh), device=dev)
    gmax = qr.grad.abs().max().item()
    bias_ref = qr.grad.sum(dim=(0, 1))                             # gradient of the packed q | k | v bias: column sums over tokens
    for fwd_out in (None, ctx):          # two passes (dQ, dK/dV) and -- bf16, S <= 192 -- the one-pass kernel that takes D from dO . ctx
        for with_bias in (False, True):
            dqkv.fill_(float("nan"))
            dbias = torch.full((3 * H,), 2.0, device=dev)           # an accumulation target: must come back as 2 + sums
            rc = L.vb_attn_bwd(code, _lib.ptr(qkv), _lib.ptr(mask_add), _lib.ptr(dctx), _lib.ptr(lse),
                               _lib.ptr(bits), _lib.ptr(ws), _lib.ptr(dqkv), _lib.ptr(fwd_out),
                               _lib.ptr(dbias) if with_bias else None, B, S, nh, 64, p, 77, 3, _lib.stream_ptr())
            _lib.check(rc, "vb_attn_bwd")
            err = (dqkv.float() - qr.grad).abs().max().item()
            if dev.type == "cuda":
                from golden_util import record
                record("attention_kernel_%s" % tag,
                       "B%d_S%d_nh%d_p%g_%s" % (B, S, nh, p, "onepass" if fwd_out is not None else "twopass"),
                       dict(ctx_err=ctx_err, dqkv_err_over_gmax=err / max(1.0, gmax), gmax=gmax))
            assert err <= pick(3e-6, ATTN_BF16_DQKV, ATTN_X3_DQKV) * max(1.0, gmax), (err, gmax, fwd_out is not None)
            if with_bias:
                berr = (dbias - 2.0 - bias_ref).abs().max().item()
                assert berr <= pick(2e-4, 0.04, 4e-4) * max(1.0, bias_ref.abs().max().item()), (berr, fwd_out is not None)


@pytest.mark.parametrize("dt", ATTN_MODES)
@pytest.mark.parametrize("cfg", [(2, 20, 36, 2, 0.0), (2, 36, 20, 2, 0.0), (1, 40, 164, 1, 0.1), (3, 7, 5, 1, 0.0),
                                 (2, 20, 100, 2, 0.1), (1, 30, 56, 1, 0.1)])   # keys at the exact 7- and 4-fragment forwards
def test_cross_attention_fwd_bwd(dev, dt, cfg):
    """queries and keys / values from different sequences of different lengths (LXRT: language <-> vision): forward and the
    two-pass backward against a torch fp32 reference; keys masked per sample; dropout keep-bits decoded and replayed."""
    from visualbert_amd import ops
    B, Sq, Sk, nh, p = cfg
    x3 = dt == "x3"
    dt, code, pick = attn_mode(dt)
    H = nh * 64
    g = torch.Generator().manual_seed(Sq * 100 + Sk)
    q = (0.7 * torch.randn(B, Sq, H, generator=g)).to(dt).to(dev)
    kv = (0.7 * torch.randn(B, Sk, 2 * H + 8, generator=g)).to(dt).to(dev)          # k | v | pad: pitches differ from H
    k, v = kv[:, :, :H], kv[:, :, H:2 * H]
    lens = torch.randint(max(Sk // 2, 1), Sk + 1, (B,), generator=g)
    mask = (torch.arange(Sk).unsqueeze(0) < lens.unsqueeze(1)).float()
    mask_add = ((1 - mask) * -10000.0).to(dev)
    L = _lib.lib()
    k2, v2, q2 = k.reshape(B * Sk, H), v.reshape(B * Sk, H), q.reshape(B * Sq, H)
    assert k2.stride(0) == 2 * H + 8
    ctx = torch.empty(B * Sq, H, dtype=dt, device=dev)
    lse = torch.empty(B, nh, Sq, device=dev)
    bits = torch.zeros(B * nh * L.vb_attn_cross_keepbits_words(Sq, Sk), dtype=torch.int64, device=dev)
    _lib.check(L.vb_attn_cross_fwd(code, _lib.ptr(q2), H, _lib.ptr(k2), k2.stride(0), _lib.ptr(v2), v2.stride(0),
                                   _lib.ptr(mask_add), _lib.ptr(ctx), H, _lib.ptr(lse), _lib.ptr(bits), B, Sq, Sk, nh, 64, p, 99, 4,
                                   _lib.stream_ptr()), "vb_attn_cross_fwd")
    keep = None
    if p > 0:
        nw = L.vb_attn_cross_keepbits_words(Sq, Sk) // (Sq * 4)
        words = bits.view(B, nh, Sq, 4, nw).cpu()
        keep = torch.zeros(B, nh, Sq, Sk)
        for key in range(Sk):
            kf, gg, r = key // 16, (key % 16) // 4, key % 4
            keep[:, :, :, key] = ((words[:, :, :, gg, kf // 16] >> ((kf % 16) * 4 + r)) & 1).float()
        keep = keep.to(dev)
        assert abs(keep.mean().item() - (1 - p)) < 0.03
    qr = q.float().detach().requires_grad_(True)
    kr = k.float().detach().clone().requires_grad_(True)
    vr = v.float().detach().clone().requires_grad_(True)

    def heads(t, S):
        return t.view(B, S, nh, 64).permute(0, 2, 1, 3)
    sc = heads(qr, Sq) @ heads(kr, Sk).transpose(-1, -2) / 8.0 + mask_add.view(B, 1, 1, Sk)
    pr = torch.softmax(sc, -1)
    if keep is not None:
        pr = pr * keep / (1 - p)
    ref = (pr @ heads(vr, Sk)).permute(0, 2, 1, 3).reshape(B, Sq, H)
    # the self-attention kernels' measured bounds (same kernels); with a handful of keys nothing averages out: the bf16 rounding of
    # the probabilities (2^-9 each) times max |v| ~ 3 is the error: 5.8e-3 measured at 5 keys
    t = pick(2e-6, 9e-3 if Sk < 16 else ATTN_BF16_CTX, 3e-5 if Sk < 16 else ATTN_X3_CTX)     # x3 at 5 keys: 1.2e-5
    assert (ctx.float().view(B, Sq, H) - ref).abs().max().item() <= t
    dctx = torch.randn(B, Sq, H, generator=g).to(dt).to(dev)
    ref.backward(dctx.float())
    dq = torch.full_like(q2, float("nan"))
    dkv = torch.full((B * Sk, 2 * H + 8), float("nan"), dtype=dt, device=dev)
    dk, dv = dkv[:, :H], dkv[:, H:2 * H]
    ws = torch.empty(B, nh, Sq, device=dev)
    d2 = dctx.reshape(B * Sq, H)
    _lib.check(L.vb_attn_cross_bwd(code, _lib.ptr(q2), H, _lib.ptr(k2), k2.stride(0), _lib.ptr(v2), v2.stride(0),
                                   _lib.ptr(mask_add), _lib.ptr(d2), H, _lib.ptr(lse), _lib.ptr(bits), _lib.ptr(ws), _lib.ptr(dq), H,
                                   _lib.ptr(dk), dk.stride(0), _lib.ptr(dv), dv.stride(0), B, Sq, Sk, nh, 64, p, 99, 4,
                                   _lib.stream_ptr()), "vb_attn_cross_bwd")
    gm = max(qr.grad.abs().max().item(), kr.grad.abs().max().item(), vr.grad.abs().max().item(), 1.0)
    lim = pick(3e-6, ATTN_BF16_DQKV, ATTN_X3_DQKV) * gm
    assert (dq.float().view(B, Sq, H) - qr.grad).abs().max().item() <= lim
    assert (dk.float().reshape(B, Sk, H) - kr.grad).abs().max().item() <= lim
    assert (dv.float().reshape(B, Sk, H) - vr.grad).abs().max().item() <= lim
    # the autograd wrapper (contiguous copies of the strided views) gives the same context
    with ops.x3_scope(x3):
        out = ops.CrossAttentionCoreFn.apply(q, k, v, mask_add, nh, 0.0, 4)
    if p == 0:
        assert (out.float() - ref).abs().max().item() <= t


@pytest.mark.parametrize("variant", [1, 14, 22, 24, 42, 80, 81, 82, 90, 92, 100, 200])     # 200: plain GEMMs through the vendor library (yardstick); 82 / 92: register-direct epilogue arms
def test_gemm_pipelined_variants_agree(dev, variant):
    """every pipelined K-contiguous kernel variant (tile shape x LDS stages, counted vmcnt, ping-pong,
    256x256) must give the generic kernel's answer -- on hardware this is what validates the
    asynchronous LDS-direct copies + counted waits (the simulator executes them synchronously)."""
    L = _lib.lib()
    M, N, K = 700, 300, 448
    g = torch.Generator().manual_seed(9)
    dt = torch.bfloat16
    A = padded(M, K, dt, dev, g)
    B = padded(N, K, dt, dev, g)
    bias = torch.randn(N, generator=g).to(dev)
    ref = A.float() @ B.float().t() + bias
    with variant_lib(dev, variant), _lib.stream_opts(nt_kernel=variant):
        for _ in range(3):
            C = gemm(dev, dt, A, B, M, N, K, 0, 0, out_f32=True, bias=bias)
            assert (C - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()


@pytest.mark.parametrize("variant", [14, 22, 24, 42, 80, 81, 82, 90, 92, 100, 200])        # 200: bias-only / "+ addend" / fp32-out go to hipBLASLt,
def test_gemm_specialised_epilogues(dev, variant):                          # the fused epilogues stay on our kernels
    """the K-contiguous fast kernels carry ONE epilogue each (activation and optional operands are template
    parameters, picked by the launcher): bias only, GELU + saved pre-activation, GELU' + fused column sums,
    residual addend -- on 16-byte aligned operands (the specialised instantiations) for bf16 -> bf16, plus the
    bf16 -> fp32 ragged-N instantiation of the MLM decoder; M ragged on purpose."""
    L = _lib.lib()
    M, N, K = 530, 512, 192
    g = torch.Generator().manual_seed(variant)
    dt = torch.bfloat16
    A = (torch.randn(M, K, generator=g) * 0.5).to(dt).to(dev)
    B = (torch.randn(N, K, generator=g) * 0.2).to(dt).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    base = A.float() @ B.float().t()
    lim = lambda ref: 1.2e-2 * max(1.0, ref.abs().max().item())
    with variant_lib(dev, variant), _lib.stream_opts(nt_kernel=variant):
        # bias only
        C = gemm(dev, dt, A, B, M, N, K, 0, 0, bias=bias)
        ref = base + bias
        assert (C.float() - ref).abs().max().item() <= lim(ref)
        # GELU + pre-activation (run-time epilogue), GELU + saved derivative (the encoder layer's forward form)
        aux = torch.zeros(M, N, dtype=dt, device=dev)
        C = gemm(dev, dt, A, B, M, N, K, 0, 0, bias=bias, act=_lib.VB_ACT_GELU, aux_out=aux)
        assert (aux.float() - ref).abs().max().item() <= lim(ref)
        assert (C.float() - torch.nn.functional.gelu(ref)).abs().max().item() <= lim(ref)
        gprime = lambda x: 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
        aux.zero_()
        C = gemm(dev, dt, A, B, M, N, K, 0, 0, bias=bias, act=_lib.VB_ACT_GELU_SAVE_GRAD, aux_out=aux)
        assert (C.float() - torch.nn.functional.gelu(ref)).abs().max().item() <= lim(ref)
        # gelu' has slope <= ~0.6: the bf16 rounding of the pre-activation it is evaluated at moves it by <= 0.6 * |ref| / 256
        assert (aux.float() - gprime(ref)).abs().max().item() <= lim(ref)
        # GELU' + column sums (run-time epilogue), multiply-by-saved-derivative + column sums (the layer's backward form)
        pre = torch.randn(M, N, generator=g).to(dt).to(dev)
        for act, factor in ((_lib.VB_ACT_GELU_GRAD, gprime(pre.float())), (_lib.VB_ACT_MUL_AUX, pre.float())):
            cs = torch.ones(N, device=dev)
            C = gemm(dev, dt, A, B, M, N, K, 0, 0, act=act, aux_in=pre, colsum=cs)
            ref2 = base * factor
            assert (C.float() - ref2).abs().max().item() <= lim(ref2)
            assert (cs - (1.0 + ref2.sum(0))).abs().max().item() <= 2e-2 * max(1.0, ref2.sum(0).abs().max().item())
        # residual addend
        add_t = torch.randn(M, N, generator=g).to(dt).to(dev)
        C = gemm(dev, dt, A, B, M, N, K, 0, 0, addend=add_t)
        ref3 = base + add_t.float()
        assert (C.float() - ref3).abs().max().item() <= lim(ref3)
        # bf16 -> fp32, ragged N (vocabulary-sized decoder): columns N-6.. are a partial 8-column group
        Nr = N - 6
        Cf = torch.full((M, N), 7.0, device=dev)[:, :Nr]
        Cf = gemm(dev, dt, A, B[:Nr], M, Nr, K, 0, 0, out_f32=True, bias=bias[:Nr].contiguous(), acc=None)
        assert (Cf - ref[:, :Nr]).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("variant", [81, 90])
def test_streaming_stores_equal_plain_stores(dev, variant):
    """The bf16 epilogues of the K <= 1024 GEMMs leave through hand-written `global_store_dwordx4 ... nt` (csrc/vb_rt.h: vb_store16_nt --
    inline asm with its own s_nop for the store-data hazard, invisible to the compiler's hazard pass).  Every epilogue
    variant of the step, on ragged M / N, for both kernels that use them: BIT-identical to the same launch with plain stores
    (developer library, debug bit 26) -- C and the auxiliary result."""
    if dev.type != "cuda":
        pytest.skip("needs libvisualbert_hip_dev.so (debug bit 26)")
    g = torch.Generator().manual_seed(260 + variant)
    dt = torch.bfloat16
    with _lib.dev_library() as L:
        L.vb_gemm_set_debug.restype, L.vb_gemm_set_debug.argtypes = _lib.DEV_SIGNATURES["vb_gemm_set_debug"]
        for (M, N, K) in ((530, 512, 192), (777, 328, 768), (256, 1024, 1024), (1000, 768, 64)):
            A = (torch.randn(M, K, generator=g) * 0.5).to(dt).to(dev)
            B = (torch.randn(N, K, generator=g) * 0.2).to(dt).to(dev)
            bias = torch.randn(N, generator=g).to(dev)
            pre = torch.randn(M, N, generator=g).to(dt).to(dev)
            add_t = torch.randn(M, N, generator=g).to(dt).to(dev)
            outs = {}
            for arm, bits in (("nt", 0), ("plain", 1 << 26)):
                L.vb_gemm_set_debug(bits)
                try:
                    with _lib.stream_opts(nt_kernel=variant):
                        res = []
                        res.append(gemm(dev, dt, A, B, M, N, K, 0, 0, bias=bias))
                        aux = torch.zeros(M, N, dtype=dt, device=dev)
                        res.append(gemm(dev, dt, A, B, M, N, K, 0, 0, bias=bias, act=_lib.VB_ACT_GELU_SAVE_GRAD, aux_out=aux))
                        res.append(aux)
                        aux2 = torch.zeros(M, N, dtype=dt, device=dev)
                        res.append(gemm(dev, dt, A, B, M, N, K, 0, 0, bias=bias, act=_lib.VB_ACT_GELU, aux_out=aux2))
                        res.append(aux2)
                        cs = torch.zeros(N, device=dev)
                        res.append(gemm(dev, dt, A, B, M, N, K, 0, 0, act=_lib.VB_ACT_MUL_AUX, aux_in=pre, colsum=cs))
                        # (the column sums themselves leave through fp32 atomics of many workgroups: not a bitwise quantity)
                        res.append(gemm(dev, dt, A, B, M, N, K, 0, 0, addend=add_t))
                        Nr = N - 6                              # ragged N: the last 8-column group is partial
                        res.append(gemm(dev, dt, A, B[:Nr], M, Nr, K, 0, 0, bias=bias[:Nr].contiguous()))
                        torch.cuda.synchronize()
                finally:
                    L.vb_gemm_set_debug(0)
                outs[arm] = res
            for i, (x, y) in enumerate(zip(outs["nt"], outs["plain"])):
                assert torch.equal(x, y), ("result %d differs between streaming and plain stores" % i, (M, N, K))
            ref = A.float() @ B.float().t() + bias                # and the streaming arm is right, not just equal
            assert (outs["nt"][0].float() - ref).abs().max().item() <= 1.2e-2 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("wgs", [0, 2])
def test_gemm_direct_b_kernel(dev, wgs):
    """nt_kernel 101: the four-wave 256x256 kernel whose B operand goes straight from global memory into MFMA-layout registers
    (1 x 4 waves, four A stages; K / 64 a multiple of 4, N a multiple of 256).  Every epilogue the step uses, a ragged M, and
    two persistent workgroups so that a workgroup walks several tiles with both operand streams running across tile boundaries."""
    M, N, K = 700, 512, 256
    g = torch.Generator().manual_seed(101)
    dt = torch.bfloat16
    A = (torch.randn(M, K, generator=g) * 0.5).to(dt).to(dev)
    B = (torch.randn(N, K, generator=g) * 0.2).to(dt).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    base = A.float() @ B.float().t()
    lim = lambda ref: 1.2e-2 * max(1.0, ref.abs().max().item())
    gprime = lambda x: 0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)
    with variant_lib(dev, 101), _lib.stream_opts(nt_kernel=101, persistent_workgroups=wgs):
        ref = base + bias
        C = gemm(dev, dt, A, B, M, N, K, 0, 0, bias=bias)
        assert (C.float() - ref).abs().max().item() <= lim(ref)
        Cf = gemm(dev, dt, A, B, M, N, K, 0, 0, out_f32=True, bias=bias)
        assert (Cf - ref).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item())
        aux = torch.zeros(M, N, dtype=dt, device=dev)
        C = gemm(dev, dt, A, B, M, N, K, 0, 0, bias=bias, act=_lib.VB_ACT_GELU_SAVE_GRAD, aux_out=aux)
        assert (C.float() - torch.nn.functional.gelu(ref)).abs().max().item() <= lim(ref)
        assert (aux.float() - gprime(ref)).abs().max().item() <= lim(ref)
        pre = torch.randn(M, N, generator=g).to(dt).to(dev)
        cs = torch.ones(N, device=dev)
        C = gemm(dev, dt, A, B, M, N, K, 0, 0, act=_lib.VB_ACT_MUL_AUX, aux_in=pre, colsum=cs)
        ref2 = base * pre.float()
        assert (C.float() - ref2).abs().max().item() <= lim(ref2)
        assert (cs - (1.0 + ref2.sum(0))).abs().max().item() <= 2e-2 * max(1.0, ref2.sum(0).abs().max().item())
        add_t = torch.randn(M, N, generator=g).to(dt).to(dev)
        C = gemm(dev, dt, A, B, M, N, K, 0, 0, addend=add_t)
        assert (C.float() - (base + add_t.float())).abs().max().item() <= lim(base)


@pytest.mark.parametrize("K", [64, 128, 192, 256, 320, 768])
def test_gemm_eight_phase_k_tails(dev, K):
    """the 8-phase 256x256 kernel has a 6-half-tile prologue and a counted-wait tail that depend on the
    number of K tiles: 1, 2, 3, 4, 5 and 12 tiles, ragged M/N edges, bf16 and fp32 outputs."""
    L = _lib.lib()
    M, N = 530, 270
    g = torch.Generator().manual_seed(K)
    dt = torch.bfloat16
    A = padded(M, K, dt, dev, g)
    B = padded(N, K, dt, dev, g)
    bias = torch.randn(N, generator=g).to(dev)
    ref = A.float() @ B.float().t() + bias
    for variant in (80, 81, 90, 100):     # eight-slot / four-slot schedules of the persistent kernel; two-workgroup kernel
        for wgs in (0, 1, 2, 4):            # 6 output tiles: one per workgroup, or 6 / 3 / 2 walked by one workgroup
            if variant in _lib.DEV_NT_KERNELS and dev.type != "cuda":
                continue                                      # no simulator build of the developer library
            with variant_lib(dev, variant), _lib.stream_opts(nt_kernel=variant, persistent_workgroups=wgs):
                for out_f32 in (True, False):
                    for _ in range(2):
                        C = gemm(dev, dt, A, B, M, N, K, 0, 0, out_f32=out_f32, bias=bias)
                        lim = (2e-3 if out_f32 else 1e-2) * ref.abs().max().item()
                        assert (C.float() - ref).abs().max().item() <= lim


def test_stream_options_do_not_leak_across_streams(dev):
    """launch options belong to a stream (vb_stream_set_opts): a GEMM on stream A keeps A's kernel choice and workgroup
    count while stream B runs with the defaults, both concurrently, and removing A's entry restores the defaults."""
    import ctypes
    L = _lib.lib()
    M, N, K = 530, 270, 192
    g = torch.Generator().manual_seed(3)
    dt = torch.bfloat16
    A = padded(M, K, dt, dev, g)
    B = padded(N, K, dt, dev, g)
    ref = A.float() @ B.float().t()
    if dev.type != "cuda":
        sa = sb = None
    else:
        sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        sa.wait_stream(torch.cuda.current_stream())
        sb.wait_stream(torch.cuda.current_stream())

    def on(stream):
        import contextlib
        return torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()

    got = _lib.StreamOpts()
    with on(sa):
        with _lib.stream_opts(nt_kernel=81, persistent_workgroups=2):
            _lib.check(L.vb_stream_get_opts(_lib.stream_ptr(), ctypes.byref(got)), "get")
            assert (got.nt_kernel, got.persistent_workgroups) == (81, 2)
            if sb is not None:
                with on(sb):                                    # the other stream sees the defaults, not A's options
                    _lib.check(L.vb_stream_get_opts(_lib.stream_ptr(), ctypes.byref(got)), "get")
                    assert (got.nt_kernel, got.persistent_workgroups, got.attn_two_pass) == (0, 0, 0)
                    Cb = gemm(dev, dt, A, B, M, N, K, 0, 0, out_f32=True)
            Ca = gemm(dev, dt, A, B, M, N, K, 0, 0, out_f32=True)
        _lib.check(L.vb_stream_get_opts(_lib.stream_ptr(), ctypes.byref(got)), "get")
        assert (got.nt_kernel, got.persistent_workgroups) == (0, 0)         # entry removed on exit
    if dev.type == "cuda":
        torch.cuda.synchronize()
        assert (Cb - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    assert (Ca - ref).abs().max().item() <= 2e-3 * ref.abs().max().item()
    bad = _lib.StreamOpts(0, 7, 0, 0)
    assert L.vb_stream_set_opts(_lib.stream_ptr(), ctypes.byref(bad)) != 0        # unknown kernel id is refused


# ---- SURVEY 8f / N4 kernels ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DTYPES)
def test_alignment_position_means(dev, dt):
    """vb_align_pos_fwd / _bwd against modeling.py:1223-1245 restated with torch ops: -1 padding, regions without any
    aligned word, an alignment tensor padded past the region count, repeated words."""
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    B, T, R, Ra, A, H, P = 3, 9, 5, 7, 4, 128, 16
    al = torch.randint(0, T, (B, Ra, A), generator=g)
    al[torch.rand(B, Ra, A, generator=g) < 0.4] = -1
    al[0, 1] = -1                                   # no aligned word at all
    al[1, 2] = 3                                    # the same word four times
    pos = torch.randn(P, H, generator=g)
    al_d, pos_d = al.to(dev), pos.to(dev)
    out = torch.empty(B * R, H, dtype=torch.float32, device=dev)
    _lib.check(L.vb_align_pos_fwd(_lib.ptr(al_d), _lib.ptr(pos_d), _lib.ptr(out), B, R, Ra, A, H, P, _lib.stream_ptr()),
               "vb_align_pos_fwd")
    pos_r = pos.clone().requires_grad_(True)
    m = (al != -1).long()
    pe = torch.nn.functional.embedding(m * al, pos_r) * m.float().unsqueeze(-1)
    cnt = m.float().sum(2)
    cnt[cnt == 0] = 1
    ref = (pe.sum(2) / cnt.unsqueeze(-1))[:, :R]
    assert (out.cpu().view(B, R, H) - ref.detach()).abs().max().item() < 1e-6
    dz = torch.randn(B, T + R, H, generator=g).to(dt)
    ref.backward(dz[:, T:].float())
    d_pos = torch.zeros(P, H, dtype=torch.float32, device=dev)
    dz_d = dz.to(dev).contiguous()
    _lib.check(L.vb_align_pos_bwd(_lib.dtype_code(dt), _lib.ptr(dz_d), _lib.ptr(al_d), _lib.ptr(d_pos), B, T, R, Ra, A, H,
                                  P, _lib.stream_ptr()), "vb_align_pos_bwd")
    assert (d_pos.cpu() - pos_r.grad).abs().max().item() < 1e-5


@pytest.mark.parametrize("dt", DTYPES)
def test_index_gather_and_scatter_rows(dev, dt):
    """batched_index_select (modeling.py:1713-1716) and its adjoint: -1 padding reads position 0, two entities on the
    same word add up, untouched rows come from the addend."""
    L = _lib.lib()
    g = torch.Generator().manual_seed(6)
    B, S, E, H = 3, 11, 5, 128
    x = torch.randn(B, S, H, generator=g).to(dt)
    idx = torch.randint(0, S, (B, E), generator=g)
    idx[0, 3:] = -1
    idx[1, 1] = idx[1, 0]
    x_d, idx_d = x.to(dev), idx.to(dev)
    out = torch.empty(B * E, H, dtype=dt, device=dev)
    _lib.check(L.vb_gather_index_rows(_lib.dtype_code(dt), _lib.ptr(x_d), _lib.ptr(idx_d), _lib.ptr(out), B, S, E, H,
                                      _lib.stream_ptr()), "vb_gather_index_rows")
    pos = idx.clamp(min=0)
    ref = x.gather(1, pos.unsqueeze(2).expand(B, E, H))
    assert torch.equal(out.cpu().view(B, E, H), ref)                     # pure data movement: bit-exact
    dsel = torch.randn(B * E, H, generator=g).to(dt)
    add = torch.randn(B * S, H, generator=g).to(dt)
    dx = torch.empty(B * S, H, dtype=dt, device=dev)
    dsel_d, add_d = dsel.to(dev), add.to(dev)               # named: a temporary would be freed before the launch reads it
    _lib.check(L.vb_scatter_index_rows(_lib.dtype_code(dt), _lib.ptr(dsel_d), _lib.ptr(idx_d), _lib.ptr(add_d),
                                       _lib.ptr(dx), B, S, E, H, _lib.stream_ptr()), "vb_scatter_index_rows")
    refx = add.float().view(B, S, H).clone()
    refx.scatter_add_(1, pos.unsqueeze(2).expand(B, E, H), dsel.float().view(B, E, H))
    assert (dx.float().cpu().view(B, S, H) - refx).abs().max().item() <= tol(dt, 1e-6, 0.02 * refx.abs().max().item())


@pytest.mark.parametrize("dt", DTYPES)
def test_flickr_scores_and_gradients(dev, dt):
    """vb_flickr_scores_fwd / _bwd against FlickrAttention.forward (modeling.py:1624-1648) and
    compute_score_with_logits_flickr (:1650-1673) written with torch ops."""
    L = _lib.lib()
    g = torch.Generator().manual_seed(7)
    B, E, R, T, d = 3, 4, 7, 6, 64
    S = T + R
    q = torch.randn(B * E, d, generator=g).to(dt)
    k = torch.randn(B * S, d, generator=g).to(dt)
    im = (torch.arange(R).unsqueeze(0) < torch.tensor([[7], [4], [5]])).long()
    lab = (torch.rand(B, E, R, generator=g) < 0.3).float() * im.unsqueeze(1).float()
    lab = lab / lab.sum(-1, keepdim=True).clamp(min=1)
    position = torch.randint(0, T, (B, E), generator=g)
    position[2, 2:] = -1
    lab[2, 2:] = 0
    scores = torch.empty(B * E, R, dtype=torch.float32, device=dev)
    stats = torch.empty(3, dtype=torch.float32, device=dev)
    q_d, k_d = q.to(dev), k.to(dev)
    im_d, lab_d, pos_d = im.to(dev), lab.contiguous().to(dev), position.to(dev)
    _lib.check(L.vb_flickr_scores_fwd(_lib.dtype_code(dt), _lib.ptr(q_d), d, _lib.ptr(k_d), d, _lib.ptr(im_d),
                                      _lib.ptr(lab_d), _lib.ptr(pos_d), _lib.ptr(scores), _lib.ptr(stats),
                                      B, E, R, S, T, d, _lib.stream_ptr()), "vb_flickr_scores_fwd")
    qr = q.float().view(B, E, d).requires_grad_(True)
    kr = k.float().view(B, S, d).requires_grad_(True)
    ref = torch.matmul(qr, kr[:, T:].transpose(-1, -2)) / math.sqrt(d) + ((1.0 - im.float()) * -10000.0).unsqueeze(1)
    assert (scores.cpu().view(B, E, R) - ref.detach()).abs().max().item() < 1e-4
    am = ref.argmax(-1, keepdim=True)
    hits = torch.gather((lab != 0).float(), 2, am).sum()
    assert stats.cpu().tolist() == [float(hits), pytest.approx(float(lab.sum()), abs=1e-5),
                                   float((position != -1).sum())]
    ds = torch.randn(B, E, R, generator=g)
    ref.backward(ds * 0.5 * 3.0)
    up = torch.tensor([0.5], device=dev)
    dq = torch.empty_like(q_d)
    dk = torch.empty_like(k_d)
    ds_d = ds.to(dev).contiguous()
    _lib.check(L.vb_flickr_scores_bwd(_lib.dtype_code(dt), _lib.ptr(ds_d), _lib.ptr(q_d), d,
                                      _lib.ptr(k_d), d, _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(up), 3.0,
                                      B, E, R, S, T, d, _lib.stream_ptr()), "vb_flickr_scores_bwd")
    eq = (dq.float().cpu().view(B, E, d) - qr.grad).abs().max().item()
    ek = (dk.float().cpu().view(B, S, d) - kr.grad).abs().max().item()
    assert eq <= tol(dt, 1e-4, 0.02 * qr.grad.abs().max().item()), eq
    assert ek <= tol(dt, 1e-4, 0.02 * kr.grad.abs().max().item()), ek
    assert float(dk.float().cpu().view(B, S, d)[:, :T].abs().max()) == 0.0     # no key gradient on the text rows


def test_small_linear_ce_over_choice_groups(dev):
    """multichoice head (modeling.py:1488-1500): Linear H -> 1 per choice, CrossEntropyLoss over logits.view(-1, 4)."""
    from visualbert_amd import ops
    g = torch.Generator().manual_seed(8)
    Bq, H = 5, 128
    x = torch.randn(Bq * 4, H, generator=g)
    w = torch.nn.Parameter(torch.randn(1, H, generator=g).to(dev))
    b = torch.nn.Parameter(torch.randn(1, generator=g).to(dev))
    lab = torch.randint(0, 4, (Bq,), generator=g)
    xd = x.to(dev).detach().requires_grad_(True)
    logits, loss = ops.SmallLinearCEFn.apply(xd, lab.to(dev), -100, w, b, 4)
    loss.backward()
    xr = x.clone().requires_grad_(True)
    wr, br = w.detach().cpu().clone().requires_grad_(True), b.detach().cpu().clone().requires_grad_(True)
    lr = torch.nn.functional.linear(xr, wr, br).view(-1, 4)
    lossr = torch.nn.functional.cross_entropy(lr, lab)
    lossr.backward()
    assert logits.shape == (Bq, 4)
    assert (logits.cpu() - lr.detach()).abs().max().item() < 1e-4
    assert abs(float(loss.detach()) - float(lossr.detach())) < 1e-5
    assert (xd.grad.cpu() - xr.grad).abs().max().item() < 1e-5
    assert (w.grad.cpu() - wr.grad).abs().max().item() < 1e-4


def test_rows_past_two_giga_elements(dev):
    """bench.py's default batch (512 x 164 = 83,968 rows x 30,528 logit columns = 2.56 G elements) indexes past int32:
    the decoder GEMM's fp32 output and the cross-entropy sweep must use 64-bit offsets.  Rows on both sides of the
    2^31-element boundary (row 70,344) are checked against torch."""
    if dev.type != "cuda":
        pytest.skip("10 GB of logits: GPU only")
    L = _lib.lib()
    g = torch.Generator().manual_seed(12)
    M, V, K = 72000, 30522, 768
    ld = 30528
    A = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16).to(dev)
    W = torch.zeros(ld, K, dtype=torch.bfloat16, device=dev)
    W[:V] = (torch.randn(V, K, generator=g) * 0.05).to(torch.bfloat16).to(dev)
    bias = torch.randn(V, generator=g).to(dev)
    C = torch.empty(M, ld, dtype=torch.float32, device=dev)
    rc = L.vb_gemm(_lib.VB_BF16, _lib.VB_F32, 0, 0, _lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(C), ld, M, V, K, 1.0, None,
                   _lib.ptr(bias), None, 0, 0, None, None, 0, 0, None, _lib.stream_ptr())
    _lib.check(rc, "vb_gemm")
    rows = torch.tensor([0, 255, 70343, 70344, 70345, 70400, 71999], device=dev)
    ref = A[rows].float() @ W[:V].float().t() + bias
    err = (C[rows, :V] - ref).abs().max().item()
    assert err <= 2e-3 * max(1.0, ref.abs().max().item()), err
    lab = torch.full((M,), -1, dtype=torch.int64, device=dev)
    picks = torch.randint(0, V, (rows.numel(),), generator=g).to(dev)
    lab[rows] = picks
    acc = torch.empty(66, device=dev); loss = torch.empty(1, device=dev)
    n = rows.numel(); n_pad = 64
    dlc = torch.full((n_pad, ld), 7.0, dtype=torch.bfloat16, device=dev)
    _lib.check(L.vb_ce_fwd_bwd_rows(_lib.VB_BF16, _lib.ptr(C), ld, _lib.ptr(lab), -1, _lib.ptr(rows), n, n_pad,
                                    _lib.ptr(acc), _lib.ptr(loss), _lib.ptr(dlc), ld, M, V, _lib.stream_ptr()),
               "vb_ce_fwd_bwd_rows")
    ref_in = C[rows, :V].detach().clone().requires_grad_(True)
    rl = torch.nn.functional.cross_entropy(ref_in, picks)
    rl.backward()
    assert abs(loss.item() - rl.item()) <= 2e-5 * max(1.0, abs(rl.item()))
    assert (dlc[:n, :V].float() - ref_in.grad).abs().max().item() <= 2e-3


def generator_keep(groups, p, seed, sid):
    """an independent numpy statement of the dropout generator documented in csrc/vb_rt.h: for each 8-element group index in
    `groups` -> bool [len(groups), 8], True = keep.  Two keyed mixer words + two words derived with a 24-bit multiply each;
    element e of a group takes 16-bit lane e of the four words; keep iff >= round(p * 65536) (clamped to 65535).  It is the
    same model whose statistics tools/dropout_rng_check.py examines."""
    import numpy as np
    M32 = np.uint64(0xFFFFFFFF)

    def u32(x):
        return x & M32

    def mix32(x, k):
        x = u32(x); x = x ^ (x >> np.uint64(16)); x = u32(x * np.uint64(0x7feb352d)); x = x ^ k
        x = x ^ (x >> np.uint64(15)); x = u32(x * np.uint64(0x846ca68b)); x = x ^ (x >> np.uint64(16))
        return x

    def mul24(a, b):
        return u32((a & np.uint64(0xFFFFFF)) * np.uint64(b & 0xFFFFFF))

    g = np.asarray(groups, dtype=np.uint64)
    s0, s1, st = np.uint64(seed & 0xFFFFFFFF), np.uint64(seed >> 32), np.uint64(sid)
    k1 = mix32(s0 ^ u32(st * np.uint64(0x9E3779B9)), s1 ^ np.uint64(0x5ca1ab1e))
    k2 = mix32(u32(s1 + u32(st * np.uint64(0x85EBCA6B))), k1) ^ u32((g >> np.uint64(31)) * np.uint64(0xC2B2AE35))
    c = u32(g * np.uint64(2) + k1)
    w0, w1 = mix32(c, k2), mix32(u32(c + np.uint64(1)), k2)
    w2, w3 = mul24(w0 >> np.uint64(8), 0x9E3779) ^ w1, mul24(w1 >> np.uint64(8), 0x85EBCB) ^ w0
    u = np.stack([h for w in (w0, w1, w2, w3) for h in (w & np.uint64(0xFFFF), w >> np.uint64(16))], 1)
    return u >= np.uint64(min(int(p * 65536.0 + 0.5), 65535))


def test_dropout_mask_equals_the_documented_generator(dev):
    """the keep mask is a pure function of (seed, stream id, element index): the elementwise kernels' masks must equal, bit for
    bit, the numpy statement of the generator above (element i = lane i%8 of group i//8)."""
    import numpy as np
    from visualbert_amd import ops
    for n, p, seed, sid in ((4099, 0.1, 1234, 10), (70000, 0.37, (7 << 32) | 99, 3)):
        x = torch.ones(n, device=dev)
        y = ops.dropout_apply(x, p, seed, sid)
        got = (y != 0).cpu().numpy()
        want = generator_keep(np.arange((n + 7) // 8), p, seed, sid).reshape(-1)[:n]
        assert np.array_equal(got, want), (n, p, int((got != want).sum()))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cfg", [(2, 37, 2, 0.1), (1, 164, 2, 0.1), (1, 112, 2, 0.1), (2, 56, 1, 0.1), (1, 300, 1, 0.25),
                                 (1, 470, 1, 0.1)])              # 470: the key-tiled forward in fp32 (absolute key positions)
def test_attention_keepbits_equal_the_documented_generator(dev, dt, cfg):
    """attention-probability dropout: the keep-bits the forward records (and the backward replays) are the same generator,
    indexed as csrc/attention.hip documents -- probability (head bh, query q, key k) is lane ((k>>4)&1)*4 + (k&3) of group
    (((bh*S + q)*4 + ((k>>2)&3)) << 5) + (k>>5).  Bit-exact for every kernel variant these shapes select (one-pass S<=192,
    the prefetching forward at 161..176, the long-sequence path above 256)."""
    import numpy as np
    B, S, nh, p = cfg
    seed, sid = (5 << 32) | 4242, 7
    H = nh * 64
    g = torch.Generator().manual_seed(21)
    qkv = (0.5 * torch.randn(B, S, 3 * H, generator=g)).to(dt).to(dev)
    mask_add = torch.zeros(B, S, device=dev)
    L = _lib.lib()
    ctx = torch.empty(B, S, H, dtype=dt, device=dev)
    lse = torch.empty(B, nh, S, device=dev)
    bits = torch.zeros(B * nh * L.vb_attn_keepbits_words(S), dtype=torch.int64, device=dev)
    _lib.check(L.vb_attn_fwd(_lib.dtype_code(dt), _lib.ptr(qkv), _lib.ptr(mask_add), _lib.ptr(ctx), _lib.ptr(lse),
                             _lib.ptr(bits), B, S, nh, 64, p, seed, sid, _lib.stream_ptr()), "vb_attn_fwd")
    got = decode_keepbits(bits, B, nh, S).view(B * nh, S, S).numpy()
    bh, q, k = np.meshgrid(np.arange(B * nh), np.arange(S), np.arange(S), indexing="ij")
    grp = ((((bh * S + q) * 4 + ((k >> 2) & 3)) << 5) + (k >> 5)).astype(np.uint64)
    lane = ((k >> 4) & 1) * 4 + (k & 3)
    want = generator_keep(grp.reshape(-1), p, seed, sid)[np.arange(grp.size), lane.reshape(-1)].reshape(got.shape)
    assert np.array_equal(got, want), int((got != want).sum())


@pytest.mark.parametrize("dt", DTYPES)
def test_head_dropout_kernel(dev, dt):
    """vb_dropout (nn.Dropout in front of the fine-tuning heads): keep rate, 1/(1-p) scaling, the same mask again for the
    same (seed, stream id) -- which is what makes it its own backward -- and a different one for another seed."""
    from visualbert_amd import ops
    n = 4099                                    # not a multiple of the 8-element generator group
    x = torch.linspace(1.0, 2.0, n).to(dt).to(dev)
    p = 0.1
    y1 = ops.dropout_apply(x, p, 1234, 10)
    y2 = ops.dropout_apply(x, p, 1234, 10)
    y3 = ops.dropout_apply(x, p, 1235, 10)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)
    keep = y1 != 0
    assert abs(keep.float().mean().item() - (1 - p)) < 0.02
    ref = (x.float() / (1 - p)).to(dt)
    assert (y1[keep].float() - ref[keep].float()).abs().max().item() <= (1e-6 if dt == torch.float32 else 0.016)
    xr = x.clone().requires_grad_(True)
    out = ops.DropoutFn.apply(xr, p, 10)
    out.float().sum().backward()
    assert torch.equal(xr.grad != 0, out != 0)                   # gradient flows exactly where the forward kept
