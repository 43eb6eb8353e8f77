#!/usr/bin/env python3
"""Device time of every single step of a bench-like run (HIP event pairs on the main stream around each step, all streams joined at
the step's end by the optimizer): does the step time drift after the warm-up steps?  python tools/step_times.py [steps]"""
import os
import sys
import time

import torch

import _variant  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hybrid_ctunet_amd as H  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    torch.manual_seed(0)
    model = H.build_model("ctunet").cuda()
    flat = H.FlatParams(H.gradient_ready_order(model))
    opt = H.FusedAdamW(None, lr=1e-4, weight_decay=1e-5, flat=flat, overlap=True)
    x, y = H.synthetic_batch(2)
    x, y = x.cuda(), y.cuda()
    loss_fn = H.LOSSES["ctunet"]
    evs, host = [], []

    def step():
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        h0 = time.perf_counter()
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = loss_fn(model(x), y)
        loss.backward()
        opt.step()
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        evs.append((e0, e1))
        host.append((time.perf_counter() - h0) * 1e3)

    for i in range(n):
        step()
        if i == 4:
            opt.freeze_skip_ranges()
    torch.cuda.synchronize()
    dev = [a.elapsed_time(b) for a, b in evs]
    gap = [evs[i][1].elapsed_time(evs[i + 1][0]) for i in range(n - 1)]
    for i in range(n):
        print(f"step {i:3d}: device {dev[i]:7.2f} ms  host enqueue {host[i]:7.2f} ms  gap to next {gap[i] if i < n - 1 else 0.0:6.2f} ms")
    print(f"mean of steps 5..24: {sum(dev[5:25]) / 20:.2f} ms (+ gaps {sum(gap[5:25]) / 20:.2f}); mean of the last 10: {sum(dev[-10:]) / 10:.2f} ms")


if __name__ == "__main__":
    main()
