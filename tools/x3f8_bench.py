#!/usr/bin/env python
"""PROTOTYPE measurement: the split-operand GEMM as it ships (three bf16 MFMA passes over hi | lo planes,
vb_gemm(VB_BF16X3) -> the persistent 256x256 kernel for these plain epilogues) against the same kernel with its two cross terms on the
block-scaled fp8 pipe (vb_gemm_x3f8, developer library): same operands, same shapes (the encoder's GEMMs at the bench's M = 167 936),
HIP-event time per launch, and both results against the fp64 product on a row sample.

    python tools/x3f8_bench.py [--rows 167936] > profiles/r04_x3f8_prototype.txt"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from visualbert_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1024 * 164)
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    L = _lib.use_dev_library()
    for name in ("vb_split_f8", "vb_gemm_x3f8"):
        fn = getattr(L, name)
        fn.restype, fn.argtypes = _lib.DEV_SIGNATURES[name]
    M = args.rows
    g = torch.Generator().manual_seed(0)
    print("# M = %d; us per launch (HIP events, %d launches after 3 warm-up); error = max |C - fp64 product| / max|product| over 64 sampled rows" % (M, args.iters))
    print("%-34s %10s %10s %8s %12s %12s" % ("shape", "bf16x3 us", "x3f8 us", "ratio", "err bf16x3", "err x3f8"))
    for name, N, K in (("QKV fwd       N=2304 K=768", 2304, 768), ("attn-out      N=768  K=768", 768, 768), ("FFN-in shape  N=3072 K=768", 3072, 768),
                       ("FFN-out fwd   N=768  K=3072", 768, 3072), ("QKV dgrad     N=768  K=2304", 768, 2304)):
        x = torch.randn(M, K, generator=g).to(dev)
        w = (torch.randn(N, K, generator=g) * 0.05).to(dev)
        bias = torch.randn(N, generator=g).to(dev)
        # the shipping mode's images
        xs = torch.empty(M, 2 * K, dtype=torch.bfloat16, device=dev)
        ws = torch.empty(N, 2 * K, dtype=torch.bfloat16, device=dev)
        _lib.check(L.vb_split_bf16(_lib.ptr(x), K, _lib.ptr(xs), 2 * K, M, K, _lib.stream_ptr()), "split")
        _lib.check(L.vb_split_bf16(_lib.ptr(w), K, _lib.ptr(ws), 2 * K, N, K, _lib.stream_ptr()), "split")
        # the prototype's images
        xi = torch.zeros(M, 4 * K, dtype=torch.uint8, device=dev)
        wi = torch.zeros(N, 4 * K, dtype=torch.uint8, device=dev)
        sx = [torch.zeros((M + 63) // 64 * 64, dtype=torch.uint8, device=dev) for _ in range(2)]
        sw = [torch.zeros((N + 63) // 64 * 64, dtype=torch.uint8, device=dev) for _ in range(2)]
        _lib.check(L.vb_split_f8(_lib.ptr(x), K, _lib.ptr(xi), 2 * K, M, K, _lib.ptr(sx[0]), _lib.ptr(sx[1]), _lib.stream_ptr()), "split_f8")
        _lib.check(L.vb_split_f8(_lib.ptr(w), K, _lib.ptr(wi), 2 * K, N, K, _lib.ptr(sw[0]), _lib.ptr(sw[1]), _lib.stream_ptr()), "split_f8")
        C3 = torch.empty(M, N, device=dev)
        C8 = torch.empty(M, N, device=dev)

        def run3():
            _lib.check(L.vb_gemm(_lib.VB_BF16X3, _lib.VB_F32, 0, 0, _lib.ptr(xs), 2 * K, _lib.ptr(ws), 2 * K, _lib.ptr(C3), N, M, N, K, 1.0, None,
                                 _lib.ptr(bias), None, 0, 0, None, None, 0, 0, None, _lib.stream_ptr()), "vb_gemm x3")

        def run8():
            _lib.check(L.vb_gemm_x3f8(_lib.ptr(xi), 2 * K, _lib.ptr(wi), 2 * K, _lib.ptr(C8), N, M, N, K, _lib.ptr(bias), _lib.ptr(sx[0]), _lib.ptr(sx[1]),
                                      _lib.ptr(sw[0]), _lib.ptr(sw[1]), _lib.stream_ptr()), "vb_gemm_x3f8")

        t = []
        for fn in (run3, run8):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            t.append(e0.elapsed_time(e1) / args.iters * 1e3)
        idx = torch.randint(0, M, (64,), generator=g).to(dev)
        ref = x[idx].double() @ w.double().t() + bias.double()
        sc = float(ref.abs().max())
        e3 = float((C3[idx].double() - ref).abs().max()) / sc
        e8 = float((C8[idx].double() - ref).abs().max()) / sc
        print("%-34s %10.1f %10.1f %8.3f %12.2e %12.2e" % (name, t[0], t[1], t[1] / t[0], e3, e8), flush=True)
        del x, w, xs, ws, xi, wi, C3, C8
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
