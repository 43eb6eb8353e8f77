#!/usr/bin/env python3
"""A/B: the two encoder-branch streams of CTUNet on DISJOINT sets of CUs (hipExtStreamCreateWithCUMask) instead of sharing the
whole chip.  If kernels of the two branches disturb each other (L2 / vector-memory contention: tools/corun.py), a spatial split
could beat dynamic sharing; if the chip is simply time-shared, it cannot.  Usage: python tools/cumask_ab.py <main CUs> (0 = no masks)"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CTU_NO_WGRAD_STREAM"] = "1"
import hybrid_ctunet_amd as H  # noqa: E402
from hybrid_ctunet_amd import ops  # noqa: E402
from hybrid_ctunet_amd.synthetic import synthetic_batch  # noqa: E402

n_main = int(sys.argv[1]) if len(sys.argv) > 1 else 0
interleave = len(sys.argv) > 2 and sys.argv[2] == "interleave"
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
hip = C.CDLL("libamdhip64.so")


def masked_stream(bits):
    words = (C.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value, device=dev)


main_stream = side = None
if n_main:
    if interleave:   # every fourth CU index to the side stream
        side_bits = [i for i in range(256) if i % 4 == 3][: 256 - n_main]
        main_bits = [i for i in range(256) if i not in set(side_bits)]
    else:
        main_bits, side_bits = list(range(n_main)), list(range(n_main, 256))
    main_stream, side = masked_stream(main_bits), masked_stream(side_bits)
    ops._side_streams[(dev, "branch")] = side
torch.manual_seed(0)
model = H.build_model("ctunet").to(dev)
flat = H.FlatParams(H.gradient_ready_order(model))
opt = H.FusedAdamW(None, lr=1e-4, weight_decay=1e-5, flat=flat)
x, y = synthetic_batch(2, seed=1000)
x, y = x.to(dev), y.to(dev)


def step():
    opt.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = H.ctunet_loss(model(x), y)
    loss.backward()
    opt.step()
    return loss


ctx = torch.cuda.stream(main_stream) if main_stream is not None else torch.cuda.stream(torch.cuda.current_stream())
with ctx:
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
print(f"main CUs {n_main or 256}{' (interleaved)' if interleave else ''}: {1e3 * dt:.2f} ms/step, loss {loss.item():.4f}")
