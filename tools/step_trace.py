#!/usr/bin/env python3
"""Where the step's phases begin and end on the device and on the launch thread, without a profiler in the way: HIP events at the
block boundaries of CTUNet (ops.trace_point: stem / layer1-4 / decoders on the main stream, ViT trunk / window stages / skip paths /
heads on the branch stream; forward when queued, backward through identity autograd nodes).  Prints, for the last of three traced
steps, every mark with its stream, device time and host (enqueue) time relative to the step's start.
python tools/step_trace.py [--serial]"""
import os
import sys
import time

import torch

import _variant  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hybrid_ctunet_amd as H  # noqa: E402
from hybrid_ctunet_amd import ops  # noqa: E402


def main():
    serial = "--serial" in sys.argv
    torch.manual_seed(0)
    model = H.build_model("ctunet").cuda()
    flat = H.FlatParams(H.gradient_ready_order(model))
    opt = H.FusedAdamW(None, lr=1e-4, weight_decay=1e-5, flat=flat, overlap=not serial)
    x, y = H.synthetic_batch(2)
    x, y = x.cuda(), y.cuda()
    ops.WGRAD_STREAM = not serial
    model.overlap_branches = not serial
    loss_fn = H.LOSSES["ctunet"]

    def step():
        ops.mark("step begin")
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = model(x)
            ops.mark("forward queued")
            loss = loss_fn(out, y)
        ops.mark("loss")
        loss.backward()
        ops.mark("backward queued")
        opt.step()
        ops.mark("step end")

    for _ in range(6):
        step()
    opt.freeze_skip_ranges()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    print(f"untraced: {(time.perf_counter() - t0) * 100:.2f} ms per step")
    ops.TRACE = []
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    print(f"traced:   {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per step")
    tr = ops.TRACE
    ops.TRACE = None
    begins = [i for i, m in enumerate(tr) if m[0] == "step begin"]
    seg = tr[begins[-1]:]
    e0, h0 = seg[0][2], seg[0][3]
    streams = {}
    rows = []
    for label, sid, ev, host in seg:
        k = streams.setdefault(sid, len(streams))
        rows.append((e0.elapsed_time(ev), k, label, (host - h0) * 1e3))
    print(f"{'device ms':>10s} {'host ms':>9s}  stream  mark")
    for dev_ms, k, label, host_ms in sorted(rows):
        print(f"{dev_ms:10.2f} {host_ms:9.2f}  {'main  ' if k == 0 else 'side' + str(k) + ' '}  {'    ' * k}{label}")


if __name__ == "__main__":
    main()
