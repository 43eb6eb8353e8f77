#!/usr/bin/env python3
"""HBM write / copy bandwidth reference points (torch fill / copy kernels on rotating buffers larger than the Infinity Cache)."""
import torch


def t(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


n = 442368 * 512
bufs = [torch.empty(n, dtype=torch.bfloat16, device="cuda") for _ in range(4)]
src = [torch.randn(n // 4, dtype=torch.float32, device="cuda").bfloat16().repeat(4) for _ in range(2)]
i = [0]


def fill():
    i[0] = (i[0] + 1) % 4
    bufs[i[0]].zero_()


def copy():
    i[0] = (i[0] + 1) % 4
    bufs[i[0]].copy_(src[i[0] % 2])


def scale():
    i[0] = (i[0] + 1) % 4
    torch.mul(src[i[0] % 2], 2.0, out=bufs[i[0]])


us = t(fill)
print(f"fill  {n * 2 / 1e6:.0f} MB: {us:.1f} us  {n * 2 / us / 1e3:.0f} GB/s write")
us = t(copy)
print(f"copy  {n * 2 / 1e6:.0f} MB: {us:.1f} us  {n * 4 / us / 1e3:.0f} GB/s read+write")
us = t(scale)
print(f"scale {n * 2 / 1e6:.0f} MB: {us:.1f} us  {n * 4 / us / 1e3:.0f} GB/s read+write")
