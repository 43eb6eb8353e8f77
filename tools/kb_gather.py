#!/usr/bin/env python3
"""In-grid strided-tap convolutions of CTUNet d101 (B = 2, bf16): LDS-DMA GEMM with a gathered operand against the generic
implicit-GEMM kernels (ctu_set_option("route", 131072)), both in one process, alternating.  python tools/kb_gather.py"""
import os
import sys

import torch

import _variant  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hybrid_ctunet_amd  # noqa: E402,F401
from hybrid_ctunet_amd import _lib, ops  # noqa: E402

CASES = [  # B, din, k, s, C, N, label
    (2, (12, 12, 24), (2, 2, 2), (2, 2, 2), 512, 1024, "patch conv 512->1024"),
    (2, (24, 24, 48), (2, 2, 2), (2, 2, 2), 256, 512, "patch conv 256->512"),
    (2, (48, 48, 96), (2, 2, 2), (2, 2, 2), 128, 256, "patch conv 128->256"),
    (2, (96, 96, 96), (2, 2, 1), (2, 2, 1), 64, 128, "ConvT(2,2,1) gradients 128<-64"),
    (2, (48, 48, 96), (1, 1, 1), (2, 2, 2), 128, 256, "1x1x1 s2 128->256"),
    (2, (24, 24, 48), (1, 1, 1), (2, 2, 2), 256, 512, "1x1x1 s2 256->512"),
    (2, (12, 12, 24), (1, 1, 1), (2, 2, 2), 512, 1024, "1x1x1 s2 512->1024"),
]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dt = torch.bfloat16
    torch.manual_seed(0)
    for B, din, k, s, C, N, label in CASES:
        dout = tuple((n - kk) // ss + 1 for n, kk, ss in zip(din, k, s))
        taps = k[0] * k[1] * k[2]
        M = B * dout[0] * dout[1] * dout[2]
        x = torch.randn(B, *din, C, device="cuda").to(dt)
        w = (torch.randn(taps, N, C, device="cuda") / (taps * C) ** 0.5).to(dt)
        gy = torch.randn(M, N, device="cuda").to(dt)
        out = torch.empty(M, N, device="cuda", dtype=dt)
        dw = torch.zeros(taps, N, C, device="cuda")
        g = ops._geom(B, din, dout, C, 0, N, k, s, (0, 0, 0), 0)
        sk = ops._conv_splitk(M, N, C, taps)
        ws = ops._splitk_workspace(x.device, M * N) if sk > 1 else None
        e = ops._epi(N, splitk_ws=ws, splitk=sk)
        flop = 2.0 * M * N * C * taps
        byt_nt = (x.numel() * (taps / (s[0] * s[1] * s[2])) + out.numel() + w.numel()) * 2
        r = {}
        for rep in range(2):
            for route in (0, 131072):
                _lib.call("ctu_set_option", b"route", route)
                r[("nt", route)] = timeit(lambda: ops._igemm_nt(x, None, w, out, g, e))
                r[("tn", route)] = timeit(lambda: ops._igemm_tn(gy, N, x, None, dw, g))
        _lib.call("ctu_set_option", b"route", 0)
        print(f"{label:34s} M={M:7d} K={taps * C:5d} N={N:5d} splitk={sk}: "
              f"nt {r[('nt', 131072)]:7.1f} -> {r[('nt', 0)]:7.1f} us ({flop / r[('nt', 0)] * 1e-6:6.1f} TFLOP/s, "
              f"{byt_nt / r[('nt', 0)] * 1e-6:5.2f} TB/s)   tn {r[('tn', 131072)]:7.1f} -> {r[('tn', 0)]:7.1f} us "
              f"({flop / r[('tn', 0)] * 1e-6:6.1f} TFLOP/s)", flush=True)


if __name__ == "__main__":
    main()
