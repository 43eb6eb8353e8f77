#!/usr/bin/env python3
"""Micro-benchmarks of single kernels at the shapes CTUNet d101 pf8 (B=2, bf16) actually launches.
Usage (GPU box): python tools/kbench.py [filter]   -> prints us/launch, algorithmic TFLOP/s and GB/s per case."""
import os
import sys
import time

import torch

import _variant  # noqa: F401  (CTU_LIB_VARIANT=TAG: a measurement build of the library)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hybrid_ctunet_amd  # noqa: E402,F401
from hybrid_ctunet_amd import ops  # noqa: E402
from hybrid_ctunet_amd._lib import Geom, call, dcode, ptr, stream  # noqa: E402

DT = torch.bfloat16
LAYOUT = int(os.environ.get("KB_LAYOUT", "0"))  # 1: CTU_LAYOUT_B16 operands for the InstanceNorm / halo conv cases
dev = "cuda"


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps  # us


def report(name, us, flops=0.0, bytes_=0.0):
    print(f"{name:58s} {us:9.1f} us  {flops / us / 1e6:8.1f} TFLOP/s  {bytes_ / us / 1e3:8.1f} GB/s", flush=True)


def tn(M, N, K):
    p = torch.randn(M, N, device=dev, dtype=DT)
    q = torch.randn(M, K, device=dev, dtype=DT)
    dw = torch.zeros(N, K, device=dev)
    g = ops._plain_geom(M, K, N)
    us = timeit(lambda: ops._igemm_tn(p, N, q, None, dw, g))
    report(f"igemm_tn M={M} N={N} K={K}", us, 2.0 * M * N * K, 2.0 * M * (N + K))


def nt(M, N, K):
    x = torch.randn(M, K, device=dev, dtype=DT)
    w = torch.randn(N, K, device=dev, dtype=DT)
    out = torch.empty(M, N, device=dev, dtype=DT)
    us = timeit(lambda: ops._plain_gemm(x, w, out, M, K, N))
    report(f"igemm_nt M={M} N={N} K={K}", us, 2.0 * M * N * K, 2.0 * (M * (N + K) + N * K))


def nt_model(M, N, K, variant, sets=4, B=2):
    """The in-model variants on rotating operand sets (4 x > 256 MB: nothing comes from the Infinity Cache):
    plain | stats (InstanceNorm sums in the epilogue) | dgrad (reduction-major W + residual)."""
    xs = [torch.randn(M, K, device=dev, dtype=DT) for _ in range(sets)]
    outs = [torch.empty(M, N, device=dev, dtype=DT) for _ in range(sets)]
    res = [torch.randn(M, N, device=dev, dtype=DT) for _ in range(sets)] if variant == "dgrad" else None
    w = torch.randn(N, K, device=dev, dtype=DT) if variant != "dgrad" else torch.randn(K, N, device=dev, dtype=DT)
    acc = torch.zeros(B * N * 2, device=dev, dtype=torch.float64)
    it = [0]

    def run():
        i = it[0] = (it[0] + 1) % sets
        if variant == "plain":
            ops._plain_gemm(xs[i], w, outs[i], M, K, N)
        elif variant == "stats":
            ops._plain_gemm(xs[i], w, outs[i], M, K, N, in_acc=acc, in_rows=M // B)
        else:
            ops._plain_gemm(xs[i], w, outs[i], M, K, N, w_kn=1, residual=res[i])
    us = timeit(run, reps=24)
    extra = 2.0 * M * N if variant == "dgrad" else 0.0
    report(f"igemm_nt[{variant}] M={M} N={N} K={K}", us, 2.0 * M * N * K, 2.0 * (M * (N + K) + N * K) + extra)


def inorm(B, D, H, W, C, sets=3):
    """InstanceNorm kernels on rotating tensors (beyond the Infinity Cache): forward apply (with stats from a fused conv
    this is all the forward does), backward reduce and backward apply."""
    xs = [torch.randn(B, D, H, W, C, device=dev, dtype=DT) for _ in range(sets)]
    gs = [torch.randn(B, D, H, W, C, device=dev, dtype=DT) for _ in range(sets)]
    ys = [torch.empty_like(x) for x in xs]
    S = D * H * W
    stats = torch.randn(B, C, 2, device=dev).abs() + 0.5
    sums = torch.zeros(B * C * 2, device=dev, dtype=torch.float64)
    dirty = torch.zeros(B * C * 2, device=dev, dtype=torch.float64)
    it = [0]
    nb = B * S * C * 2.0

    def nxt():
        it[0] = (it[0] + 1) % sets
        return it[0]

    def f_apply():
        i = nxt()
        call("ctu_in_apply", dcode(DT), ptr(xs[i]), ptr(stats), None, ptr(ys[i]), B, S, C, 1, LAYOUT, None, stream())

    def f_stats():
        i = nxt()
        call("ctu_in_stats", dcode(DT), ptr(xs[i]), B, S, C, ptr(sums), ptr(stats), stream())

    def f_red():
        i = nxt()
        call("ctu_in_bwd_reduce", dcode(DT), ptr(gs[i]), ptr(xs[i]), None, ptr(stats), ptr(sums), B, S, C, 1, None, stream())

    def f_bapply():
        i = nxt()
        call("ctu_in_bwd_apply", dcode(DT), ptr(gs[i]), ptr(xs[i]), None, ptr(stats), ptr(sums), ptr(ys[i]), None, B, S, C, 1,
             ptr(dirty), B * C * 2, LAYOUT, None, stream())
    sync = torch.zeros(128, device=dev, dtype=torch.int32)

    def f_pair():
        i = nxt()
        call("ctu_in_bwd_reduce", dcode(DT), ptr(gs[i]), ptr(xs[i]), None, ptr(stats), ptr(sums), B, S, C, 1, None, stream())
        call("ctu_in_bwd_apply", dcode(DT), ptr(gs[i]), ptr(xs[i]), None, ptr(stats), ptr(sums), ptr(ys[i]), None, B, S, C, 1,
             ptr(dirty), B * C * 2, LAYOUT, None, stream())

    def f_fused():
        i = nxt()
        call("ctu_in_bwd_fused", dcode(DT), ptr(gs[i]), ptr(xs[i]), None, ptr(stats), ptr(sums), ptr(ys[i]), None, B, S, C, 1,
             ptr(dirty), B * C * 2, LAYOUT, None, ptr(sync), stream())
    tag = f"{C}ch @{D}x{H}x{W} B{B}"
    report(f"in_apply       {tag}", timeit(f_apply), 0, 2 * nb)
    report(f"in_bwd_reduce  {tag}", timeit(f_red), 0, 2 * nb)
    report(f"in_bwd_apply   {tag}", timeit(f_bapply), 0, 3 * nb)
    report(f"in_bwd pair    {tag}", timeit(f_pair), 0, 5 * nb)
    if nb <= (32 << 20):
        report(f"in_bwd fused   {tag}", timeit(f_fused), 0, 5 * nb)


def halo(B, D, H, W, C, N, what):
    x = torch.randn(B, D, H, W, C, device=dev, dtype=DT)
    w = torch.nn.Parameter(torch.randn(N, C, 3, 3, 3, device=dev) * 0.05)
    flops = 2.0 * B * D * H * W * N * C * 27
    if what == "fwd":
        with torch.no_grad():
            ops.FUSE_IN_STATS = False
            if LAYOUT:
                x._ctu_b16 = True   # timing only: the same bytes read through the blocked addressing
            us = timeit(lambda: ops.conv3d(x, w, 1, 1))
            ops.FUSE_IN_STATS = True
        report(f"conv3_halo fwd {C}->{N} @{D}x{H}x{W} B{B}", us, flops, 2.0 * B * D * H * W * (C + N))
    else:
        dy = torch.randn(B, D, H, W, N, device=dev, dtype=DT)
        panel = torch.zeros(27, N, C, device=dev)
        wws = torch.empty(256 * 54 * 1024 + 27 * 64 * 1024, device=dev)
        us = timeit(lambda: call("ctu_conv3_halo_wgrad", dcode(DT), ptr(dy), ptr(x), None, ptr(panel), B, D, H, W, C, 0, N,
                                 LAYOUT, LAYOUT, ptr(wws), wws.numel(), stream()))
        report(f"conv3_halo wgrad {C}->{N} @{D}x{H}x{W} B{B}", us, flops, 2.0 * B * D * H * W * (C + N))


def ff(M, Hd):
    """Fused FeedForward forward (ctu_ff_fwd) against the three launches it replaces (LayerNorm, GEMM + GELU + pre, GEMM + residual)."""
    D = 128
    sets = 3
    xs = [torch.randn(M, D, device=dev, dtype=DT) for _ in range(sets)]
    g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    w1, b1 = torch.randn(Hd, D, device=dev, dtype=DT) * 0.09, torch.zeros(Hd, device=dev)
    w2, b2 = torch.randn(D, Hd, device=dev, dtype=DT) * 0.04, torch.zeros(D, device=dev)
    w2f = torch.empty(D * Hd, device=dev, dtype=DT)
    call("ctu_ff_pack_w2", ptr(w2), ptr(w2f), D, Hd, stream())
    ys = [torch.empty(M, D, device=dev, dtype=DT) for _ in range(sets)]
    pres = [torch.empty(M, Hd, device=dev, dtype=DT) for _ in range(sets)]
    us = [torch.empty(M, Hd, device=dev, dtype=DT) for _ in range(sets)]
    hs = [torch.empty(M, D, device=dev, dtype=DT) for _ in range(sets)]
    mr = torch.empty(M, 2, device=dev)
    it = [0]

    def fused():
        i = it[0] = (it[0] + 1) % sets
        call("ctu_ff_fwd", dcode(DT), ptr(xs[i]), ptr(g), ptr(b), ptr(w1), ptr(b1), ptr(w2f), ptr(b2), ptr(ys[i]), ptr(pres[i]), ptr(us[i]),
             ptr(mr), M, D, Hd, stream())

    def three():
        i = it[0] = (it[0] + 1) % sets
        call("ctu_layernorm_fwd", dcode(DT), ptr(xs[i]), ptr(g), ptr(b), ptr(hs[i]), ptr(mr), M, D, stream())
        ops._plain_gemm(hs[i], w1, us[i], M, D, Hd, bias=b1, act=1, pre_out=pres[i])
        ops._plain_gemm(us[i], w2, ys[i], M, Hd, D, bias=b2, residual=xs[i])
    byt = 2.0 * M * (2 * D + 2 * Hd)
    report(f"ff fused        M={M} D={D} Hd={Hd}", timeit(fused), 4.0 * M * D * Hd, byt)
    report(f"ff three launches M={M} D={D} Hd={Hd}", timeit(three), 4.0 * M * D * Hd, byt)


def pwa(M):
    """Fused cross-weight forward (ctu_pwa_block_fwd, with and without the saved projections) against the six launches it replaces."""
    C = 128
    sets = 3
    x1 = [torch.randn(M, C, device=dev, dtype=DT) for _ in range(sets)]
    x2 = [torch.randn(M, C, device=dev, dtype=DT) for _ in range(sets)]
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    wq1, wq2 = torch.randn(3 * C, C, device=dev, dtype=DT) * 0.09, torch.randn(3 * C, C, device=dev, dtype=DT) * 0.09
    wo = torch.randn(C, C, device=dev, dtype=DT) * 0.09
    wpk = torch.empty(4 * 56 * 512, device=dev, dtype=DT)
    call("ctu_pwa_pack", ptr(wq1), ptr(wq2), ptr(wo), ptr(wpk), C, stream())
    outs = [torch.empty(M, C, device=dev, dtype=DT) for _ in range(sets)]
    q1 = [torch.empty(M, 3 * C, device=dev, dtype=DT) for _ in range(sets)]
    q2 = [torch.empty(M, 3 * C, device=dev, dtype=DT) for _ in range(sets)]
    h1 = [torch.empty(M, C, device=dev, dtype=DT) for _ in range(sets)]
    h2 = [torch.empty(M, C, device=dev, dtype=DT) for _ in range(sets)]
    oo = [torch.empty(M, C, device=dev, dtype=DT) for _ in range(sets)]
    mr1, mr2 = torch.empty(M, 2, device=dev), torch.empty(M, 2, device=dev)
    it = [0]
    scale = 32 ** -0.5

    def fused(save):
        i = it[0] = (it[0] + 1) % sets
        call("ctu_pwa_block_fwd", dcode(DT), ptr(x1[i]), ptr(x2[i]), ptr(g), ptr(b), ptr(g), ptr(b), ptr(wpk), ptr(outs[i]),
             ptr(q1[i]) if save else None, ptr(q2[i]) if save else None, ptr(mr1), ptr(mr2), M, C, scale, stream())

    def six():
        i = it[0] = (it[0] + 1) % sets
        call("ctu_layernorm_fwd", dcode(DT), ptr(x1[i]), ptr(g), ptr(b), ptr(h1[i]), ptr(mr1), M, C, stream())
        ops._plain_gemm(h1[i], wq1, q1[i], M, C, 3 * C)
        call("ctu_layernorm_fwd", dcode(DT), ptr(x2[i]), ptr(g), ptr(b), ptr(h2[i]), ptr(mr2), M, C, stream())
        ops._plain_gemm(h2[i], wq2, q2[i], M, C, 3 * C)
        call("ctu_pwa_fwd", dcode(DT), ptr(q1[i]), ptr(q2[i]), ptr(oo[i]), M, C, scale, stream())
        ops._plain_gemm(oo[i], wo, outs[i], M, C, C)
    flop = 2.0 * M * C * (6 * C + C)
    report(f"pwa fused, projections saved M={M}", timeit(lambda: fused(True)), flop, 2.0 * M * (3 * C + 6 * C))
    report(f"pwa fused, nothing saved     M={M}", timeit(lambda: fused(False)), flop, 2.0 * M * 3 * C)
    report(f"pwa six launches             M={M}", timeit(six), flop, 2.0 * M * (3 * C + 6 * C))


def colsum(M, N, scaled):
    x = torch.randn(M, N, device=dev, dtype=DT)
    rs = torch.randn(M, device=dev, dtype=DT) if scaled else None
    out = torch.zeros(N, device=dev)
    us = timeit(lambda: call("ctu_colsum", dcode(DT), ptr(x), ptr(rs), M, N, N, ptr(out), stream()))
    report(f"colsum M={M} N={N} scaled={scaled}", us, 0, 2.0 * M * N)


CASES = {
    "colsum": lambda: [colsum(1769472, 64, True), colsum(1769472, 64, True), colsum(1769472, 16, False), colsum(442368, 128, False)],
    "ff": lambda: [ff(442368, 512), ff(55296, 512)],
    "pwa": lambda: [pwa(442368), pwa(55296)],
    "tn_trunk": lambda: [tn(864, 3072, 768), tn(864, 768, 3072), tn(864, 2304, 768), tn(864, 768, 768)],
    "tn_big": lambda: [tn(442368, 512, 128), tn(442368, 128, 512), tn(442368, 384, 128), tn(442368, 128, 32),
                       tn(442368, 32, 128), tn(55296, 768, 256), tn(55296, 256, 64), tn(1769472, 16, 64),
                       tn(6912, 1536, 512)],
    "tn_skinny": lambda: [tn(55296, 256, 64), tn(55296, 64, 256), tn(442368, 128, 32), tn(442368, 32, 128),
                          tn(1769472, 16, 64), tn(6912, 512, 128), tn(6912, 128, 512)],
    "tn_probe": lambda: [tn(13824, 256, 64), tn(55296, 256, 64), tn(221184, 256, 64), tn(55296, 64, 64),
                         tn(55296, 128, 64), tn(55296, 256, 32), tn(55296, 256, 128), tn(55296, 64, 32)],
    "nt_trunk": lambda: [nt(864, 768, 3072), nt(864, 3072, 768), nt(864, 768, 768), nt(864, 2304, 768)],
    "nt_big": lambda: [nt(442368, 512, 128), nt(442368, 128, 512), nt(442368, 128, 32), nt(442368, 32, 128),
                       nt(6912, 128, 512), nt(1769472, 16, 64), nt(55296, 256, 64)],
    "nt_model": lambda: [nt_model(442368, 512, 128, v) for v in ("plain", "stats", "dgrad")] +
                        [nt_model(442368, 128, 32, v) for v in ("plain", "stats")] +
                        [nt_model(442368, 32, 128, v) for v in ("plain", "dgrad")] +
                        [nt_model(442368, 384, 128, "plain"), nt_model(442368, 128, 512, "plain"),
                         nt_model(442368, 128, 384, "dgrad"), nt_model(55296, 1024, 256, "stats"),
                         nt_model(55296, 256, 64, "stats"), nt_model(55296, 64, 256, "dgrad")],
    "halo_debug": lambda: [(call("ctu_set_option", b"nt_debug", d), print("nt_debug =", d),
                            halo(2, 96, 96, 96, 64, 64, "fwd"), halo(2, 48, 48, 96, 128, 128, "fwd"),
                            halo(2, 24, 24, 48, 256, 256, "fwd"), call("ctu_set_option", b"nt_debug", 0)) for d in (0, 4, 8, 12)],
    "inorm": lambda: [inorm(2, 96, 96, 96, 64), inorm(2, 48, 48, 96, 128), inorm(2, 48, 48, 96, 512), inorm(2, 24, 24, 48, 256),
                      inorm(2, 24, 24, 48, 1024), inorm(2, 48, 48, 96, 32)],
    "inorm_small": lambda: [inorm(2, 12, 12, 24, 128, sets=8), inorm(2, 12, 12, 24, 512, sets=8), inorm(2, 24, 24, 48, 64, sets=8),
                            inorm(2, 24, 24, 48, 256, sets=8), inorm(2, 6, 6, 12, 256, sets=8), inorm(2, 6, 6, 12, 1024, sets=8),
                            inorm(2, 48, 48, 96, 32, sets=6)],
    "wgrad_debug": lambda: [(call("ctu_set_option", b"nt_debug", d), print("nt_debug =", d),
                             halo(2, 96, 96, 96, 64, 64, "wgrad"), halo(2, 48, 48, 96, 128, 128, "wgrad"),
                             halo(2, 24, 24, 48, 256, 256, "wgrad"), call("ctu_set_option", b"nt_debug", 0)) for d in (0, 4, 1, 5)],
    "nt_small": lambda: [nt_model(55296, 256, 64, "stats", sets=8), nt_model(55296, 256, 64, "plain", sets=8),
                         nt_model(6912, 512, 128, "stats", sets=16), nt_model(6912, 512, 128, "plain", sets=16),
                         nt_model(6912, 128, 128, "plain", sets=16), nt_model(55296, 128, 64, "plain", sets=8)],
    "nt_debug": lambda: [(call("ctu_set_option", b"nt_debug", d), print("nt_debug =", d),
                          nt_model(442368, 512, 128, "plain"), nt_model(442368, 128, 32, "plain"),
                          nt_model(55296, 1024, 256, "plain"), call("ctu_set_option", b"nt_debug", 0)) for d in (0, 1, 2, 3)],
    "halo_fwd": lambda: [halo(2, 96, 96, 96, 64, 64, "fwd"), halo(2, 48, 48, 96, 128, 128, "fwd"),
                         halo(2, 24, 24, 48, 256, 256, "fwd"), halo(2, 48, 48, 96, 32, 32, "fwd"),
                         halo(2, 12, 12, 24, 512, 512, "fwd")],
    "halo_small": lambda: [halo(2, 12, 12, 24, 128, 128, "fwd"), halo(2, 6, 6, 12, 256, 256, "fwd"),
                           halo(2, 24, 24, 48, 64, 64, "fwd"), halo(2, 12, 12, 24, 512, 512, "fwd"),
                           halo(2, 12, 12, 24, 128, 128, "wgrad"), halo(2, 6, 6, 12, 256, 256, "wgrad"),
                           halo(2, 24, 24, 48, 64, 64, "wgrad"), halo(2, 12, 12, 24, 512, 512, "wgrad")],
    # tail quantisation: 1 728 bricks (3.375 rounds of 512 resident workgroups) against 1 536 (3 rounds) and 2 048 (4 rounds)
    "halo_rounds": lambda: [halo(2, 48, 48, 96, 128, 128, "fwd"), halo(2, 48, 32, 128, 128, 128, "fwd"), halo(2, 64, 32, 128, 128, 128, "fwd"),
                            halo(2, 48, 48, 96, 64, 64, "fwd"), halo(2, 96, 96, 96, 64, 64, "fwd")],
    "halo_wgrad": lambda: [halo(2, 96, 96, 96, 64, 64, "wgrad"), halo(2, 48, 48, 96, 128, 128, "wgrad"),
                           halo(2, 24, 24, 48, 256, 256, "wgrad"), halo(2, 48, 48, 96, 32, 32, "wgrad")],
}

if os.environ.get("KB_ROUTE"):
    call("ctu_set_option", b"route", int(os.environ["KB_ROUTE"]))

if __name__ == "__main__":
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    for k, fn in CASES.items():
        if flt in k:
            fn()
