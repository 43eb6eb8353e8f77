#!/usr/bin/env python3
"""Every entry-point launch of one training step by (entry point, shape arguments): count, total and average HIP-event time.
Single stream, per-op host path (the per-launch profiler switches the launch lists off), CTUNet d101 pf8, B = 2, bf16.
python tools/shape_table.py [name filter] [top]"""
import ctypes
import os
import sys

import torch

import _variant  # noqa: F401

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hybrid_ctunet_amd as H  # noqa: E402
from hybrid_ctunet_amd import _lib, ops  # noqa: E402


class Prof:
    def __init__(self):
        self.names = set(_lib._SIGS) - {"ctu_set_option", "ctu_plan_run", "ctu_plan_create", "ctu_plan_destroy"}
        self.rec = []
        self.on = False

    def add(self, name, args, e0, e1):
        if not self.on:
            return
        key = [name[4:]]
        for a in args:
            if isinstance(a, ctypes.Structure):
                key.append("{" + ",".join(str(getattr(a, f)) for f, _ in a._fields_ if isinstance(getattr(a, f), int) and f not in
                                          ("bias", "residual", "out2", "splitk_ws", "in_acc", "pre_out")) + "}")
            elif isinstance(a, bool):
                key.append(str(int(a)))
            elif isinstance(a, int):
                key.append(str(a) if abs(a) < (1 << 28) else "p")
            elif a is None:
                key.append("-")
            elif isinstance(a, float):
                key.append(f"{a:.3g}")
            else:
                key.append("?")
        self.rec.append((" ".join(key), e0, e1))


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    torch.manual_seed(0)
    model = H.build_model("ctunet").cuda()
    flat = H.FlatParams(H.gradient_ready_order(model))
    opt = H.FusedAdamW(None, lr=1e-4, weight_decay=1e-5, flat=flat, overlap=False)
    x, y = H.synthetic_batch(2)
    x, y = x.cuda(), y.cuda()
    ops.WGRAD_STREAM = False
    model.overlap_branches = False
    model.enc0_stream = False
    prof = Prof()
    loss_fn = H.LOSSES["ctunet"]

    def step():
        opt.zero_grad()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = loss_fn(model(x), y)
        loss.backward()
        opt.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    _lib.PROFILER = prof
    step()                 # per-op path warm-up (its own workspaces / packed panels)
    prof.on = True
    step()
    torch.cuda.synchronize()
    _lib.PROFILER = None
    agg = {}
    for key, e0, e1 in prof.rec:
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += e0.elapsed_time(e1) * 1e3
    tot = sum(v[1] for v in agg.values())
    print(f"# {len(prof.rec)} launches, {tot / 1e3:.2f} ms of event time (single stream, per-op path, one step)")
    fam = {}
    for k, v in agg.items():
        f = fam.setdefault(k.split()[0], [0, 0.0])
        f[0] += v[0]
        f[1] += v[1]
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print(f"## {k:28s} x{v[0]:5d} {v[1] / 1e3:8.3f} ms")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        if flt in k:
            print(f"{v[1]:9.1f} us  x{v[0]:3d}  avg {v[1] / v[0]:8.1f} us  {k}")


if __name__ == "__main__":
    main()
