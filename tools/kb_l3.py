#!/usr/bin/env python3
"""Does the order of launches decide whether a re-read comes from the 256 MiB Infinity Cache?  
InstanceNorm backward = reduce (reads dy, x) then apply (reads dy, x again, writes dx); forward = conv (writes y) then apply
(reads y, writes a).  Each pair is timed whole-batch (B = 2: the working set between the two reads of a line is the whole
tensor pair) and per batch item (B = 1 launches, item 0 then item 1: half the working set).  Operands rotate over `sets`
independent tensor sets so nothing survives from the previous repetition."""
import os
import sys

import torch

import _variant  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybrid_ctunet_amd import ops  # noqa: E402
from hybrid_ctunet_amd._lib import call, dcode, ptr, stream  # noqa: E402

DT = torch.bfloat16
dev = "cuda"


def timeit(fn, reps=12, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def bwd_pair(B, D, H, W, C, sets=4):
    S = D * H * W
    xs = [torch.randn(B, S, C, device=dev, dtype=DT) for _ in range(sets)]
    gs = [torch.randn(B, S, C, device=dev, dtype=DT) for _ in range(sets)]
    ys = [torch.empty(B, S, C, device=dev, dtype=DT) for _ in range(sets)]
    stats = torch.rand(B, C, 2, device=dev) + 0.5
    sums = torch.zeros(B * C * 2, device=dev, dtype=torch.float64)
    dirty = torch.zeros(B * C * 2, device=dev, dtype=torch.float64)
    it = [0]

    def red(i, b0, nb):
        call("ctu_in_bwd_reduce", dcode(DT), ptr(gs[i][b0]), ptr(xs[i][b0]), None, ptr(stats[b0]), ptr(sums[b0 * C * 2:]), nb, S, C, 1, None,
             stream())

    def app(i, b0, nb):
        call("ctu_in_bwd_apply", dcode(DT), ptr(gs[i][b0]), ptr(xs[i][b0]), None, ptr(stats[b0]), ptr(sums[b0 * C * 2:]), ptr(ys[i][b0]), None,
             nb, S, C, 1, ptr(dirty[b0 * C * 2:]), nb * C * 2, 0, None, stream())

    def whole():
        i = it[0] = (it[0] + 1) % sets
        red(i, 0, B)
        app(i, 0, B)

    def per_item():
        i = it[0] = (it[0] + 1) % sets
        for b in range(B):
            red(i, b, 1)
            app(i, b, 1)

    def only_red():
        i = it[0] = (it[0] + 1) % sets
        red(i, 0, B)

    def only_app():
        i = it[0] = (it[0] + 1) % sets
        app(i, 0, B)

    tag = f"{C}ch @{D}x{H}x{W} B{B} ({B * S * C * 2 / 1e6:.0f} MB per tensor)"
    tr, ta = timeit(only_red), timeit(only_app)
    tw, tp = timeit(whole), timeit(per_item)
    print(f"IN backward {tag}: reduce {tr:7.1f} us, apply {ta:7.1f} us alone; reduce->apply whole batch {tw:7.1f} us, "
          f"per item {tp:7.1f} us ({100 * (tp - tw) / tw:+.1f} %)", flush=True)


def fwd_pair(B, D, H, W, C, N, sets=3):
    """conv3_halo (fused statistics) -> in_apply_acc, whole batch vs per item."""
    S = D * H * W
    xs = [torch.randn(B, D, H, W, C, device=dev, dtype=DT) for _ in range(sets)]
    ys = [torch.empty(B, D, H, W, N, device=dev, dtype=DT) for _ in range(sets)]
    outs = [torch.empty(B, D, H, W, N, device=dev, dtype=DT) for _ in range(sets)]
    w = torch.nn.Parameter(torch.randn(N, C, 3, 3, 3, device=dev) * 0.05)
    with torch.no_grad():
        wfr = ops._pack_frag(w, N, C, 27, C * 27, 27, 1, 0, DT)
    acc = torch.zeros(B * N * 2, device=dev, dtype=torch.float64)
    other = torch.zeros(B * N * 2, device=dev, dtype=torch.float64)
    stats = torch.empty(B, N, 2, device=dev)
    ws = ops._tn_workspace(xs[0].device)
    it = [0]

    def conv(i, b0, nb):
        call("ctu_conv3_halo", dcode(DT), ptr(xs[i][b0]), None, ptr(wfr), ptr(ys[i][b0]), None, nb, D, H, W, C, 0, N, 0, N, 0,
             ptr(acc[b0 * N * 2:]), None, None, ptr(ws), ws.numel(), 0, stream())

    def app(i, b0, nb):
        call("ctu_in_apply_acc", dcode(DT), ptr(ys[i][b0]), ptr(acc[b0 * N * 2:]), ptr(stats[b0]), None, ptr(outs[i][b0]), nb, S, N, 1, 0,
             None, ptr(other[b0 * N * 2:]), nb * N * 2, stream())

    def whole():
        i = it[0] = (it[0] + 1) % sets
        conv(i, 0, B)
        app(i, 0, B)

    def per_item():
        i = it[0] = (it[0] + 1) % sets
        for b in range(B):
            conv(i, b, 1)
            app(i, b, 1)

    def only_conv():
        i = it[0] = (it[0] + 1) % sets
        conv(i, 0, B)

    def only_app():
        i = it[0] = (it[0] + 1) % sets
        app(i, 0, B)

    tc, ta = timeit(only_conv), timeit(only_app)
    tw, tp = timeit(whole), timeit(per_item)
    print(f"forward {C}->{N} @{D}x{H}x{W} B{B}: conv {tc:7.1f} us, apply {ta:7.1f} us alone; conv->apply whole batch {tw:7.1f} us, "
          f"per item {tp:7.1f} us ({100 * (tp - tw) / tw:+.1f} %)", flush=True)


if __name__ == "__main__":
    for shp in [(2, 48, 48, 96, 128), (2, 96, 96, 96, 64), (2, 48, 48, 96, 32), (2, 24, 24, 48, 256), (2, 96, 96, 96, 32)]:
        bwd_pair(*shp)
    for shp in [(2, 48, 48, 96, 128, 128), (2, 96, 96, 96, 64, 64)]:
        fwd_pair(*shp)
