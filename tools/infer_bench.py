#!/usr/bin/env python3
"""Timed sliding-window inference (reference: trainer_CTUNet.py:417-557 as called by val_epoch_hybrid,
:188-190 - roi 96^3, sw_batch_size 4, overlap 0.5) of CTUNet d101 pf8 on one synthetic CT-sized volume: windows per second,
peak memory during the pass, and how much of that memory is activations kept for a backward pass that never comes.

The model runs in eval mode under torch.no_grad() and autocast(bf16), the way the reference's validation loop calls it.  Under
no_grad every fused block (ops_fused.py) drops its intermediates when its forward returns: the same forward with gradients
enabled is run once beside it to show the difference."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hybrid_ctunet_amd as H  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = H.build_model("ctunet").to(dev).eval()
D, Hh, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) >= 4 else (192, 192, 160)))
vol = torch.rand(1, 1, D, Hh, W, device=dev)
starts = H.inference.window_starts((D, Hh, W), (96, 96, 96), 0.5)
n_win = len(starts)


def run():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        return H.sliding_window_inference(vol, (96, 96, 96), 4, model, overlap=0.5, mode="gaussian")


for _ in range(2):
    out = run()
torch.cuda.synchronize()
torch.cuda.reset_peak_memory_stats()
base = torch.cuda.memory_allocated()
reps = 3
per_rep = []
t0 = time.perf_counter()
for _ in range(reps):
    t1 = time.perf_counter()
    out = run()
    torch.cuda.synchronize()
    per_rep.append(time.perf_counter() - t1)
dt = (time.perf_counter() - t0) / reps
print("passes: " + " ".join(f"{1e3 * t:.1f}" for t in per_rep) + " ms")
peak = torch.cuda.max_memory_allocated() - base
assert all(torch.isfinite(o).all() for o in out)
print(f"volume {D}x{Hh}x{W}: {n_win} windows of 96^3 (overlap 0.5, gaussian blend), sw_batch_size 4")
print(f"sliding_window_inference: {1e3 * dt:.1f} ms per volume = {n_win / dt:.1f} windows/s "
      f"({1e3 * dt / ((n_win + 3) // 4):.1f} ms per forward of 4 windows)")
print(f"peak memory above the resident model during the pass: {peak / 2**30:.2f} GiB")
# the same 4-window forward with gradients enabled (what a training-mode forward keeps for backward)
x4 = torch.rand(4, 1, 96, 96, 96, device=dev)
for grad in (False, True):
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    b0 = torch.cuda.memory_allocated()
    with torch.set_grad_enabled(grad), torch.autocast("cuda", dtype=torch.bfloat16):
        o = model(x4)
    torch.cuda.synchronize()
    held = torch.cuda.memory_allocated() - b0
    pk = torch.cuda.max_memory_allocated() - b0
    print(f"one forward of 4 windows, grad {'on ' if grad else 'off'}: held afterwards {held / 2**30:6.2f} GiB, peak {pk / 2**30:6.2f} GiB")
    del o
