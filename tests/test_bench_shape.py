"""Parity at the shape bench.py TIMES ("Next round" 1): per-GPU batch 1024 x (128 tokens + 36 regions) =
167,936 token rows, bf16 -- 656-row-tile GEMMs, 12,288 attention workgroups, [1024][3H] bias-partial slots, and a
[167936, 30528] fp32 logits tensor of 5.13 G elements (past 2^32).  No golden of the reference can exist at this size
(the fp32 CPU reference would need days), so every stage of one BertLayer forward + backward is checked against a torch
fp32 recompute FROM THE KERNELS' OWN STAGE INPUTS (torch's hipBLASLt / ATen kernels are an independent implementation);
the chain of per-stage checks covers the whole layer, over EVERY row -- not a sample.

  test_bert_layer_at_bench_shape      vb_bert_layer_fwd / vb_bert_layer_bwd, B = 1024, S = 164, ragged masks, p = 0
  test_bert_layer_dropout_run_is_deterministic   the same call with p = 0.1 twice: bit-identical activations and input
                                      gradients (a race in an asynchronous copy pipeline shows up as a flipped bit)
  test_logits_past_four_giga_elements decoder GEMM + vb_ce_fwd_bwd_rows on both sides of the 2^31- and 2^32-element marks
  test_race_screen_at_bench_rows      tools/race_screen.py (every NT kernel variant + the grouped wgrad kernel against fp32
                                      torch matmuls, workgroup counts that make a workgroup walk many tiles) with the bench's M
Reference lines replaced: pytorch_pretrained_bert/modeling.py:231-341 (BertLayer), :417-420 + :1471-1473 (decoder + loss)."""
import ctypes
import os
import subprocess
import sys

import pytest
import torch

from visualbert_amd import _lib

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BF = torch.bfloat16


class Dims:
    def __init__(self, B, S, H, I, NH):
        self.B, self.S, self.H, self.I, self.NH, self.M = B, S, H, I, NH, B * S


BENCH = Dims(1024, 164, 768, 3072, 12)       # bench.py's default per-GPU batch (configs[1])
SMALL = Dims(3, 164, 128, 256, 2)            # the same checks at a size the kernel-logic simulator finishes (VB_EMU=1)


def _al(x):
    return (x + 255) & ~255


def _carve(buf, specs):
    """views into a byte buffer laid out like csrc/layer.hip's carve_saved / carve_scratch: consecutive 256-byte-aligned
    regions; specs = [(name, dtype, shape)]"""
    out, o = {}, 0
    for name, dt, shape in specs:
        n = 1
        for s in shape:
            n *= s
        nbytes = n * torch.empty(0, dtype=dt).element_size()
        out[name] = buf[o:o + nbytes].view(dt).view(*shape)
        o += _al(nbytes)
    return out, o


def _saved_views(D, saved, keep_words):
    B, S, H, I, NH, M = D.B, D.S, D.H, D.I, D.NH, D.M
    specs = [("qkv", BF, (M, 3 * H)), ("ctx", BF, (M, H)), ("z1", BF, (M, H)), ("a_out", BF, (M, H)), ("pre", BF, (M, I)),
             ("inter", BF, (M, I)), ("z2", BF, (M, H)), ("lse", torch.float32, (B, NH, S)), ("mean1", torch.float32, (M,)),
             ("rstd1", torch.float32, (M,)), ("mean2", torch.float32, (M,)), ("rstd2", torch.float32, (M,)),
             ("keepbits", torch.int64, (keep_words,))]
    v, total = _carve(saved, specs)
    assert total == saved.numel(), (total, saved.numel())     # the layout mirrored here IS the library's
    return v


def _scratch_views(D, scratch):
    H, I, M = D.H, D.I, D.M
    specs = [("t_h%d" % i, BF, (M, H)) for i in range(6)] + [("t_i", BF, (M, I)), ("t_3h", BF, (M, 3 * H))]
    return _carve(scratch, specs)[0]


def _close(got, ref, rel, floor_of_max, what):
    """elementwise |got - ref| <= rel |ref| + floor_of_max max|ref|, in row chunks (no full-size temporaries kept)"""
    scale = float(ref.abs().max())
    assert scale > 0 and scale == scale, what
    worst = 0.0
    for r0 in range(0, got.size(0), 32768):
        g_, r_ = got[r0:r0 + 32768].float(), ref[r0:r0 + 32768].float()
        excess = (g_ - r_).abs() - rel * r_.abs()
        worst = max(worst, float(excess.max()))
    assert worst <= floor_of_max * scale, "%s: excess error %.4g over %.4g (max|ref| %.4g)" % (what, worst, floor_of_max * scale, scale)


def _ptr_array(items):
    arr = (ctypes.c_void_p * len(items))()
    for i, t in enumerate(items):
        arr[i] = t.data_ptr() if t is not None else None
    return arr


def _layer_problem(D, dev, seed=5):
    B, S, H, I, M = D.B, D.S, D.H, D.I, D.M
    g = torch.Generator(device=dev).manual_seed(seed)

    def rn(*shape, scale=1.0, dt=torch.float32):
        return (torch.randn(*shape, generator=g, device=dev) * scale).to(dt)

    P = dict(h_in=rn(M, H, dt=BF), d_out=rn(M, H, scale=0.05, dt=BF),
             wqkv=rn(3 * H, H, scale=0.04, dt=BF), bqkv=rn(3 * H, scale=0.1), wo=rn(H, H, scale=0.04, dt=BF), bo=rn(H, scale=0.1),
             g1=1.0 + rn(H, scale=0.1), b1=rn(H, scale=0.1), wi=rn(I, H, scale=0.04, dt=BF), bi=rn(I, scale=0.1),
             wo2=rn(H, I, scale=0.02, dt=BF), bo2=rn(H, scale=0.1), g2=1.0 + rn(H, scale=0.1), b2=rn(H, scale=0.1))
    # ragged batch: text padded in the MIDDLE of the sequence (slots T' .. 127), regions at the tail (128 + R' .. 163)
    T, R = 128, 36
    tl = torch.randint(T // 2, T + 1, (B,), generator=g, device=dev)
    rl = torch.randint(R // 2, R + 1, (B,), generator=g, device=dev)
    tl[-1], rl[-1] = T, R                                                  # the last sample is full
    pos = torch.arange(S, device=dev)[None, :]
    valid = torch.where(pos < T, pos < tl[:, None], (pos - T) < rl[:, None])
    P["mask_add"] = ((~valid).float() * -10000.0).contiguous()
    for k in ("wqkv", "wo", "wi", "wo2"):                                  # W^T shadows [in, out]
        P[k + "_t"] = P[k].t().contiguous()
    return P


def _run_layer(D, P, dev, p_hidden, p_attn, seed=0x1234567, sid=40):
    B, S, H, I, NH, M = D.B, D.S, D.H, D.I, D.NH, D.M
    L = _lib.lib()
    code = _lib.VB_BF16
    nsaved = L.vb_bert_layer_saved_bytes(code, B, S, H, I, NH, float(p_attn))
    nscr = L.vb_bert_layer_scratch_bytes(code, B, S, H, I, NH)
    saved = torch.empty(nsaved, dtype=torch.uint8, device=dev)
    scratch = torch.empty(nscr, dtype=torch.uint8, device=dev)
    h_out = torch.empty(M, H, dtype=BF, device=dev)
    weights = [P["wqkv"], P["bqkv"], P["wo"], P["bo"], P["g1"], P["b1"], P["wi"], P["bi"], P["wo2"], P["bo2"], P["g2"], P["b2"]]
    _lib.check(L.vb_bert_layer_fwd(code, _lib.ptr(P["h_in"]), _lib.ptr(P["mask_add"]), _lib.ptr(h_out), _lib.ptr(saved),
                                   _lib.ptr(scratch), _ptr_array(weights), B, S, H, I, NH, p_hidden, p_attn, 1e-12, seed, sid,
                                   _lib.stream_ptr()), "vb_bert_layer_fwd")
    return saved, scratch, h_out, weights


def _run_bwd(D, P, saved, scratch, weights, dev, p_hidden, p_attn, seed=0x1234567, sid=40):
    B, S, H, I, NH, M = D.B, D.S, D.H, D.I, D.NH, D.M
    L = _lib.lib()
    grads = [torch.zeros(w.shape, dtype=torch.float32, device=dev) for w in weights]
    d_in = torch.empty(M, H, dtype=BF, device=dev)
    wts = [P["wqkv_t"], P["wo_t"], P["wi_t"], P["wo2_t"]]
    ld_t = (ctypes.c_int64 * 4)(*[w.stride(0) for w in wts])
    _lib.check(L.vb_bert_layer_bwd(_lib.VB_BF16, _lib.ptr(P["h_in"]), _lib.ptr(P["mask_add"]), _lib.ptr(P["d_out"]), _lib.ptr(d_in),
                                   _lib.ptr(saved), _lib.ptr(scratch), _ptr_array(weights), _ptr_array(grads), _ptr_array(wts),
                                   ld_t, B, S, H, I, NH, p_hidden, p_attn, seed, sid, _lib.stream_ptr()), "vb_bert_layer_bwd")
    return d_in, grads


def _ln_ref(z, gamma, beta):
    mean = z.mean(-1, keepdim=True)
    var = ((z - mean) ** 2).mean(-1, keepdim=True)
    return gamma * (z - mean) / torch.sqrt(var + 1e-12) + beta


def _ln_bwd_ref(dy, z, mean, rstd, gamma):
    xhat = (z - mean[:, None]) * rstd[:, None]
    dxh = dy * gamma
    return rstd[:, None] * (dxh - dxh.mean(-1, keepdim=True) - xhat * (dxh * xhat).mean(-1, keepdim=True)), xhat


def _attn_ref(D, qkv, mask_add, dctx=None, chunk=64):
    B, S, H, NH, M = D.B, D.S, D.H, D.NH, D.M
    """softmax(QK^T/8 + mask) V per (sample, head) in fp32, from the kernel's own packed qkv; with dctx also dqkv"""
    ctx = torch.empty(M, H, dtype=torch.float32, device=qkv.device)
    dqkv = torch.empty(M, 3 * H, dtype=torch.float32, device=qkv.device) if dctx is not None else None
    for b0 in range(0, B, chunk):
        nb = min(chunk, B - b0)
        x = qkv[b0 * S:(b0 + nb) * S].float().view(nb, S, 3, NH, 64).permute(2, 0, 3, 1, 4).contiguous()
        x.requires_grad_(dctx is not None)
        q, k, v = x[0], x[1], x[2]
        sc = q @ k.transpose(-1, -2) / 8.0 + mask_add[b0:b0 + nb, None, None, :]
        c = torch.softmax(sc, -1) @ v                                   # [nb, nh, S, 64]
        ctx[b0 * S:(b0 + nb) * S] = c.detach().permute(0, 2, 1, 3).reshape(nb * S, H)
        if dctx is not None:
            dc = dctx[b0 * S:(b0 + nb) * S].float().view(nb, S, NH, 64).permute(0, 2, 1, 3)
            c.backward(dc)
            dqkv[b0 * S:(b0 + nb) * S] = x.grad.permute(1, 3, 0, 2, 4).reshape(nb * S, 3 * H)
            x.grad = None
    return ctx, dqkv


def _sized(dev, which):
    if which == "bench" and dev.type != "cuda":
        pytest.skip("bench-sized layer: GPU only")
    if which == "small" and dev.type == "cuda" and os.environ.get("VB_SMALL_ON_GPU") != "1":
        pytest.skip("the small size validates this test's own references on the kernel-logic simulator (VB_EMU=1)")
    return BENCH if which == "bench" else SMALL


@pytest.mark.parametrize("which", ["bench", "small"])
def test_bert_layer_at_bench_shape(dev, which):
    D = _sized(dev, which)
    H = D.H
    P = _layer_problem(D, dev)
    saved, scratch, h_out, weights = _run_layer(D, P, dev, 0.0, 0.0)
    sv, sc = _saved_views(D, saved, 0), _scratch_views(D, scratch)
    f = lambda t: t.float()
    # ---- forward, stage by stage, every row
    _close(sv["qkv"], f(P["h_in"]) @ f(P["wqkv"]).t() + P["bqkv"], 0.005, 0.004, "qkv = h_in Wqkv^T + b")
    ctx_ref, _ = _attn_ref(D, sv["qkv"], P["mask_add"])
    _close(sv["ctx"], ctx_ref, 0.01, 0.01, "attention context")
    del ctx_ref
    ao = sc["t_h0"]
    _close(ao, f(sv["ctx"]) @ f(P["wo"]).t() + P["bo"], 0.005, 0.004, "attention-out dense")
    z1 = f(ao) + f(P["h_in"])
    _close(sv["z1"], z1, 0.004, 0.001, "z1 = attention-out + residual")               # one fp32 add, one rounding
    _close(sv["a_out"], _ln_ref(z1, P["g1"], P["b1"]), 0.005, 0.004, "LayerNorm 1")
    del z1
    x = f(sv["a_out"]) @ f(P["wi"]).t() + P["bi"]
    _close(sv["inter"], torch.nn.functional.gelu(x), 0.005, 0.004, "FFN-in + erf GELU")
    cdf = 0.5 * (1.0 + torch.erf(x * 0.70710678118654752440))
    _close(sv["pre"], cdf + x * torch.exp(-0.5 * x * x) * 0.39894228040143267794, 0.005, 0.004, "saved GELU'")
    del x, cdf
    fo = sc["t_h1"]
    _close(fo, f(sv["inter"]) @ f(P["wo2"]).t() + P["bo2"], 0.005, 0.004, "FFN-out dense")
    z2 = f(fo) + f(sv["a_out"])
    _close(sv["z2"], z2, 0.004, 0.001, "z2 = FFN-out + residual")
    _close(h_out, _ln_ref(z2, P["g2"], P["b2"]), 0.005, 0.004, "LayerNorm 2 (h_out)")
    del z2
    # ---- backward (reuses the scratch: the forward temporaries above are dead from here)
    d_in, G = _run_bwd(D, P, saved, scratch, weights, dev, 0.0, 0.0)
    QKV_W, QKV_B, AO_W, AO_B, LN1_G, LN1_B, FI_W, FI_B, FO_W, FO_B, LN2_G, LN2_B = range(12)
    dz2_ref, xhat2 = _ln_bwd_ref(f(P["d_out"]), f(sv["z2"]), sv["mean2"], sv["rstd2"], P["g2"])
    dz2 = sc["t_h0"]
    _close(dz2, dz2_ref, 0.01, 0.004, "LayerNorm 2 backward")

    def vec_close(got, ref, tol, what):
        assert float((got - ref).abs().max()) <= tol * float(ref.abs().max()), (what, float((got - ref).abs().max()), float(ref.abs().max()))

    vec_close(G[LN2_G], (f(P["d_out"]) * xhat2).sum(0), 2e-3, "d gamma 2")
    vec_close(G[LN2_B], f(P["d_out"]).sum(0), 2e-3, "d beta 2")
    vec_close(G[FO_B], f(dz2).sum(0), 1e-2, "FFN-out bias gradient")
    del dz2_ref, xhat2
    dpre = sc["t_i"]
    _close(dpre, (f(dz2) @ f(P["wo2"])) * f(sv["pre"]), 0.005, 0.004, "dgrad FFN-out x GELU'")
    vec_close(G[FI_B], f(dpre).sum(0), 1e-2, "FFN-in bias gradient (fused column sums)")
    da = sc["t_h2"]
    _close(da, f(dpre) @ f(P["wi"]) + f(dz2), 0.005, 0.004, "dgrad FFN-in + residual gradient")
    dz1_ref, xhat1 = _ln_bwd_ref(f(da), f(sv["z1"]), sv["mean1"], sv["rstd1"], P["g1"])
    dz1 = sc["t_h5"]
    _close(dz1, dz1_ref, 0.01, 0.004, "LayerNorm 1 backward")
    vec_close(G[LN1_G], (f(da) * xhat1).sum(0), 2e-3, "d gamma 1")
    vec_close(G[AO_B], f(dz1).sum(0), 1e-2, "attention-out bias gradient")
    del dz1_ref, xhat1
    dctx = sc["t_h3"]
    _close(dctx, f(dz1) @ f(P["wo"]), 0.005, 0.004, "dgrad attention-out")
    dqkv = sc["t_3h"]
    _, dqkv_ref = _attn_ref(D, sv["qkv"], P["mask_add"], dctx)
    _close(dqkv, dqkv_ref, 0.02, 0.01, "attention backward (dqkv): one workgroup per (sample, head)")
    vec_close(G[QKV_B], dqkv_ref.sum(0), 1e-2, "q|k|v bias gradient from the one-pass kernel's accumulators")
    del dqkv_ref
    _close(d_in, f(dqkv) @ f(P["wqkv"]) + f(dz1), 0.005, 0.004, "dgrad QKV + residual gradient (d_in)")
    # the four weight gradients of the grouped launch (fp32 atomics over token slices)
    for idx, dy, xx, what in ((FO_W, dz2, sv["inter"], "dW FFN-out"), (FI_W, dpre, sv["a_out"], "dW FFN-in"),
                              (AO_W, dz1, sv["ctx"], "dW attention-out"), (QKV_W, dqkv, P["h_in"], "dW QKV")):
        ref = f(dy).t() @ f(xx)
        vec_close(G[idx], ref, 1e-3, what)
        del ref
    assert H == D.H


@pytest.mark.parametrize("which", ["bench", "small"])
def test_bert_layer_dropout_run_is_deterministic(dev, which):
    """what the bench times: p_hidden = p_attn = 0.1.  The masks are a pure function of (seed, site, element), every kernel
    but the atomically accumulated weight gradients is order-independent: two runs must agree bit for bit."""
    D = _sized(dev, which)
    P = _layer_problem(D, dev, seed=6)
    outs = []
    for rep in range(2):
        saved, scratch, h_out, weights = _run_layer(D, P, dev, 0.1, 0.1)
        kw = _lib.lib().vb_attn_keepbits_words(D.S) * D.B * D.NH
        sv = _saved_views(D, saved, kw)
        d_in, G = _run_bwd(D, P, saved, scratch, weights, dev, 0.1, 0.1)
        sc = _scratch_views(D, scratch)
        outs.append([h_out.clone(), sv["ctx"].clone(), sv["keepbits"].clone(), d_in.clone(), sc["t_3h"].clone(), G[1].clone()])
        if rep == 0:
            assert torch.isfinite(h_out.float()).all() and torch.isfinite(d_in.float()).all()
    for a, b, what in zip(outs[0], outs[1], ("h_out", "ctx", "keep-bits", "d_in", "dqkv")):
        assert torch.equal(a, b), what
    # the q|k|v bias gradient: per-workgroup partial sums, then a second stage whose summation order is not fixed -- equal to
    # fp32 round-off, not bit for bit (measured r03: 4 significant digits printed identical, torch.equal false)
    a, b = outs[0][5], outs[1][5]
    assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max())
    _, _, h0, _ = _run_layer(D, P, dev, 0.0, 0.0)
    assert not torch.equal(h0, outs[0][0])                                # dropout really ran


def test_logits_past_four_giga_elements(dev):
    """bench.py's default batch: 167,936 rows x 30,528 logit columns = 5.13 G fp32 elements (20.5 GB).  The decoder GEMM's
    output offsets and the cross-entropy sweep must be 64-bit: rows on both sides of the 2^31-element mark (row 70,344) and
    of the 2^32-element mark (row 140,689 holds it) are checked against torch, plus the first and last tiles."""
    if dev.type != "cuda":
        pytest.skip("20 GB of logits: GPU only")
    L = _lib.lib()
    g = torch.Generator().manual_seed(12)
    M = BENCH.M
    V, K, ld = 30522, 768, 30528
    assert (140689 * ld) < 2 ** 32 < (140690 * ld)
    A = (torch.randn(M, K, generator=g) * 0.5).to(BF).to(dev)
    W = torch.zeros(ld, K, dtype=BF, device=dev)
    W[:V] = (torch.randn(V, K, generator=g) * 0.05).to(BF).to(dev)
    bias = torch.randn(V, generator=g).to(dev)
    C = torch.empty(M, ld, dtype=torch.float32, device=dev)
    _lib.check(L.vb_gemm(_lib.VB_BF16, _lib.VB_F32, 0, 0, _lib.ptr(A), K, _lib.ptr(W), K, _lib.ptr(C), ld, M, V, K, 1.0, None,
                         _lib.ptr(bias), None, 0, 0, None, None, 0, 0, None, _lib.stream_ptr()), "vb_gemm")
    rows = torch.tensor([0, 255, 256, 70343, 70344, 70345, 131071, 131072, 140688, 140689, 140690, 140691, 167679, 167680,
                         167935], device=dev)
    ref = A[rows].float() @ W[:V].float().t() + bias
    err = (C[rows, :V] - ref).abs().max().item()
    assert err <= 2e-3 * max(1.0, ref.abs().max().item()), err
    # every 997th row as well (a stride co-prime with every tile size): 169 rows spread over all 656 row tiles' neighbourhood
    rows2 = torch.arange(0, M, 997, device=dev)
    ref2 = A[rows2].float() @ W[:V].float().t() + bias
    assert (C[rows2, :V] - ref2).abs().max().item() <= 2e-3 * max(1.0, ref2.abs().max().item())
    lab = torch.full((M,), -1, dtype=torch.int64, device=dev)
    picks = torch.randint(0, V, (rows.numel(),), generator=g).to(dev)
    lab[rows] = picks
    acc = torch.empty(66, device=dev)
    loss = torch.empty(1, device=dev)
    n, n_pad = rows.numel(), 64
    dlc = torch.full((n_pad, ld), 7.0, dtype=BF, device=dev)
    _lib.check(L.vb_ce_fwd_bwd_rows(_lib.VB_BF16, _lib.ptr(C), ld, _lib.ptr(lab), -1, _lib.ptr(rows), n, n_pad,
                                    _lib.ptr(acc), _lib.ptr(loss), _lib.ptr(dlc), ld, M, V, _lib.stream_ptr()),
               "vb_ce_fwd_bwd_rows")
    ref_in = C[rows, :V].detach().clone().requires_grad_(True)
    rl = torch.nn.functional.cross_entropy(ref_in, picks)
    rl.backward()
    assert abs(loss.item() - rl.item()) <= 2e-5 * max(1.0, abs(rl.item()))
    assert (dlc[:n, :V].float() - ref_in.grad).abs().max().item() <= 2e-3
    assert float(dlc[n:].float().abs().max()) == 0.0                     # padding rows of the compact gradient are zeroed


def test_race_screen_at_bench_rows(dev):
    """tools/race_screen.py behind pytest: all K-contiguous kernel variants (22 / 42 / 80 / 81 / 90 / 100, full chip and
    reduced workgroup counts so a persistent workgroup walks many tiles) and the grouped wgrad kernel against fp32 torch,
    including the bench's own M = 167,936 rows.  Needs the developer build of the library (kernel selection knobs)."""
    if dev.type != "cuda":
        pytest.skip("GPU only")
    if not os.path.isfile(os.path.join(ROOT, "visualbert_amd", "libvisualbert_hip_dev.so")):
        pytest.skip("developer library not built")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "race_screen.py"), "1", "--bench-rows"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout, r.stdout[-2000:]
