#!/usr/bin/env python3
"""Where a workgroup of ctu_pwa_block_fwd spends its time: shader-clock stamps of the first two tiles of every wave (variant
library built with -DPW_STAMPS: tools/build_variant.sh stamps pwa_fused.hip -DPW_STAMPS; CTU_LIB_VARIANT=stamps)."""
import os
import sys

import torch

import _variant  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybrid_ctunet_amd._lib import call, dcode, ptr, stream  # noqa: E402

M, C = 442368, 128
dev, DT = "cuda", torch.bfloat16
x1, x2 = torch.randn(M, C, device=dev, dtype=DT), torch.randn(M, C, device=dev, dtype=DT)
g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
wq1, wq2 = torch.randn(3 * C, C, device=dev, dtype=DT) * 0.09, torch.randn(3 * C, C, device=dev, dtype=DT) * 0.09
wo = torch.randn(C, C, device=dev, dtype=DT) * 0.09
wpk = torch.empty(4 * 56 * 512, device=dev, dtype=DT)
call("ctu_pwa_pack", ptr(wq1), ptr(wq2), ptr(wo), ptr(wpk), C, stream())
out = torch.empty(M, C, device=dev, dtype=DT)
st = torch.zeros(256 * 4 * 64, device=dev, dtype=torch.int64)
q2 = torch.empty(8, device=dev, dtype=DT)
mr1, mr2 = torch.empty(M, 2, device=dev), torch.empty(M, 2, device=dev)
for _ in range(3):
    call("ctu_pwa_block_fwd", dcode(DT), ptr(x1), ptr(x2), ptr(g), ptr(b), ptr(g), ptr(b), ptr(wpk), ptr(out),
         ptr(st), ptr(q2), ptr(mr1), ptr(mr2), M, C, 32 ** -0.5, stream())
torch.cuda.synchronize()
t = st.view(256, 4, 64).cpu()
names = ["tile start", "rows loaded, LayerNorms"]
for h in range(4):
    names += [f"h{h} stage landed", f"h{h} computed"]
names += ["rows staged"]
n = len(names)
for tile in range(3):
    d = (t[:, :, tile * n + 1:tile * n + n] - t[:, :, tile * n:tile * n + n - 1]).double()
    print(f"tile {tile}: mean shader-clock cycles per phase over 256 workgroups x 4 waves (min .. max)")
    for i in range(n - 1):
        print(f"   {names[i + 1]:22s} {d[:, :, i].mean():9.0f}   ({d[:, :, i].min():7.0f} .. {d[:, :, i].max():7.0f})")
    tot = (t[:, :, tile * n + n - 1] - t[:, :, tile * n]).double()
    print(f"   {'whole tile':22s} {tot.mean():9.0f}")
