#!/usr/bin/env python3
"""How long does the host need to ENQUEUE one training step (no device sync inside the loop)?  If this approaches the
GPU step time, launches become the limiter and HIP-graph capture is due."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hybrid_ctunet_amd as H
from hybrid_ctunet_amd.synthetic import synthetic_batch

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = H.build_model("ctunet").to(dev)
flat = H.FlatParams(H.gradient_ready_order(model))
opt = H.FusedAdamW(None, lr=1e-4, weight_decay=1e-5, flat=flat)
x, y = synthetic_batch(2, seed=1000)
x, y = x.to(dev), y.to(dev)

def step():
    opt.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = H.LOSSES["ctunet"](model(x), y)
    loss.backward()
    opt.step()

for _ in range(3):
    step()
torch.cuda.synchronize()
n = 10
t0 = time.time()
for _ in range(n):
    step()
t1 = time.time()
torch.cuda.synchronize()
t2 = time.time()
# (with a launch thread faster than the device this first figure is NOT the host's cost: the HIP queues fill up and the launch
#  calls block until the device drains them - the empty-queue figures below are what the host itself needs)
print(f"back-to-back steps: loop returns after {1e3 * (t1 - t0) / n:.1f} ms/step (queue back-pressure included), "
      f"device-complete {1e3 * (t2 - t0) / n:.1f} ms/step")
# one step enqueued into an EMPTY queue, forward and backward separately: host time without back-pressure
for _ in range(3):
    torch.cuda.synchronize()
    a = time.time()
    opt.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = H.LOSSES["ctunet"](model(x), y)
    b = time.time()
    torch.cuda.synchronize()
    c = time.time()
    loss.backward()
    opt.step()
    d = time.time()
    torch.cuda.synchronize()
    e = time.time()
    print(f"forward: host {1e3 * (b - a):.1f} ms (device done after {1e3 * (c - a):.1f}); "
          f"backward+opt: host {1e3 * (d - c):.1f} ms (device done after {1e3 * (e - c):.1f})")
