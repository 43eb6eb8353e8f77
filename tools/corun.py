#!/usr/bin/env python3
"""Do two kernels of the step make progress side by side, or do they take turns?  For pairs (A, B) of the step's large kernels
- MFMA-bound halo convolution forward / weight gradient, HBM-bound InstanceNorm backward apply, a 442 k-row GEMM - time n
launches of A alone, m launches of B alone (n, m chosen so both queues take about the same time) and both queues together on
two HIP streams.  overlap = (tA + tB - tAB) / min(tA, tB): 1 = the shorter queue is hidden entirely, 0 = strictly serial."""
import os
import sys

import torch

import _variant  # noqa: F401  (CTU_LIB_VARIANT=TAG: a measurement build of the library)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hybrid_ctunet_amd  # noqa: E402,F401
from hybrid_ctunet_amd import ops  # noqa: E402
from hybrid_ctunet_amd._lib import call, dcode, ptr, stream  # noqa: E402

DT = torch.bfloat16
dev = "cuda"
B, D, H, W, C = 2, 96, 96, 96, 64
x = torch.randn(B, D, H, W, C, device=dev, dtype=DT)
dy = torch.randn(B, D, H, W, C, device=dev, dtype=DT)
out = torch.empty_like(x)
w = torch.nn.Parameter(torch.randn(C, C, 3, 3, 3, device=dev) * 0.05)
S = D * H * W
stats = torch.rand(B, C, 2, device=dev) + 0.5
sums = torch.zeros(B * C * 2, device=dev, dtype=torch.float64)
dirty = torch.zeros(B * C * 2, device=dev, dtype=torch.float64)
M = 442368
ga = torch.randn(M, 128, device=dev, dtype=DT)
gw = torch.randn(512, 128, device=dev, dtype=DT)
go = torch.empty(M, 512, device=dev, dtype=DT)
gdw = torch.zeros(512, 128, device=dev)
wws = torch.empty(256 * 54 * 1024 + 27 * 64 * 1024, device=dev)
panel = torch.zeros(27, C, C, device=dev)
with torch.no_grad():
    wfr = ops._pack_frag(w, C, C, 27, C * 27, 27, 1, 0, DT)
tnws = {}


def k_halo():
    ws = ops._tn_workspace(x.device)
    call("ctu_conv3_halo", dcode(DT), ptr(x), None, ptr(wfr), ptr(out), None, B, D, H, W, C, 0, C, 0, C, 0, None, None, None,
         ptr(ws), ws.numel(), 0, stream())


def k_wgrad():
    call("ctu_conv3_halo_wgrad", dcode(DT), ptr(dy), ptr(x), None, ptr(panel), B, D, H, W, C, 0, C, 0, 0, ptr(wws), wws.numel(),
         stream())


def k_inbwd():
    call("ctu_in_bwd_apply", dcode(DT), ptr(dy), ptr(x), None, ptr(stats), ptr(sums), ptr(out), None, B, S, C, 1, ptr(dirty),
         B * C * 2, 0, None, stream())


def k_inred():
    call("ctu_in_bwd_reduce", dcode(DT), ptr(dy), ptr(x), None, ptr(stats), ptr(sums), B, S, C, 1, None, stream())


def k_gemm():
    ops._plain_gemm(ga, gw, go, M, 128, 512)


def k_tn():
    ops._igemm_tn(go, 512, ga, None, gdw, ops._plain_geom(M, 128, 512))


K = {"halo_fwd": k_halo, "halo_wgrad": k_wgrad, "in_bwd_apply": k_inbwd, "in_bwd_reduce": k_inred, "gemm_nt": k_gemm, "gemm_tn": k_tn}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(fa, na, fb, nb):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    s1.wait_stream(torch.cuda.current_stream())
    s2.wait_stream(torch.cuda.current_stream())
    # interleave the enqueues so neither queue starves on the host side
    ia = ib = 0
    while ia < na or ib < nb:
        if ia < na:
            with torch.cuda.stream(s1):
                fa()
            ia += 1
        if ib < nb:
            with torch.cuda.stream(s2):
                fb()
            ib += 1
    torch.cuda.current_stream().wait_stream(s1)
    torch.cuda.current_stream().wait_stream(s2)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3


if os.environ.get("KB_ROUTE"):
    call("ctu_set_option", b"route", int(os.environ["KB_ROUTE"]))
alone = {}
for n, f in K.items():
    with torch.cuda.stream(s1):   # workspaces of both streams exist before timing
        f()
    with torch.cuda.stream(s2):
        f()
    run(f, 3, f, 0)
    alone[n] = run(f, 20, f, 0) / 20
    print(f"{n:14s} alone {alone[n]:7.1f} us")
pairs = [("halo_fwd", "in_bwd_apply"), ("halo_fwd", "gemm_nt"), ("halo_fwd", "halo_wgrad"), ("halo_wgrad", "in_bwd_apply"),
         ("halo_wgrad", "gemm_nt"), ("halo_wgrad", "gemm_tn"), ("in_bwd_apply", "gemm_nt"), ("halo_fwd", "halo_fwd"),
         ("in_bwd_apply", "in_bwd_reduce")]
for a, b in pairs:
    na = 12
    nb = max(1, round(na * alone[a] / alone[b]))
    ta, tb = alone[a] * na, alone[b] * nb
    run(K[a], 2, K[b], 2)
    tab = run(K[a], na, K[b], nb)
    print(f"{a:13s} x{na:3d} ({ta:7.0f} us) || {b:13s} x{nb:3d} ({tb:7.0f} us): together {tab:7.0f} us   "
          f"overlap {(ta + tb - tab) / min(ta, tb):5.2f}")
