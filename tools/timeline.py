#!/usr/bin/env python3
"""Concurrency picture of one profiled run: python tools/timeline.py <kernel_trace.csv> [skip_fraction | marker:first:steps:per_step]
(marker form: the window runs from the start of launch number first * per_step of the kernel whose name contains `marker` to the
start of launch (first + steps) * per_step - whole steps, e.g. im2col_cin1:4:5:2 = five steps from the fifth on)
Reads rocprofv3 --kernel-trace CSV (Start_Timestamp / End_Timestamp / Queue_Id / Kernel_Name), drops the first
skip_fraction of the time span (warm-up) and prints: wall span, busy time (union of kernel intervals), time with
1 / 2 / 3+ kernels in flight, idle gaps, and per kernel name the time it ran ALONE (nothing else on the device)."""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
win = sys.argv[2] if len(sys.argv) > 2 else "0.45"
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
if ":" in win:
    marker, first, nsteps, per = win.split(":")
    marks = [r[0] for r in rows if marker in r[2]]
    cut, end = marks[int(first) * int(per)], marks[(int(first) + int(nsteps)) * int(per)]
    rows = [r for r in rows if cut <= r[0] < end]
    t0, t1 = cut, end
    print(f"window: {nsteps} steps, {(t1 - t0) / 1e6 / int(nsteps):.2f} ms per step")
else:
    cut = t0 + int((t1 - t0) * float(win))
    rows = [r for r in rows if r[0] >= cut]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
events = []
for i, (s, e, n, q) in enumerate(rows):
    events.append((s, 1, i))
    events.append((e, -1, i))
events.sort()
active = set()
last = t0
hist = defaultdict(int)
alone = defaultdict(int)
for t, d, i in events:
    if t > last:
        k = len(active)
        hist[min(k, 3)] += t - last
        if k == 1:
            alone[rows[next(iter(active))][2]] += t - last
        last = t
    if d > 0:
        active.add(i)
    else:
        active.discard(i)
span = t1 - t0
print(f"span {span / 1e6:.2f} ms, kernels {len(rows)}, queues {len({r[3] for r in rows})}")
for k in (0, 1, 2, 3):
    print(f"  {k}{'+' if k == 3 else ' '} kernels in flight: {hist[k] / 1e6:8.2f} ms  ({100.0 * hist[k] / span:5.1f} %)")
tot_alone = sum(alone.values())
print(f"time with exactly one kernel in flight, by kernel (top 25 of {tot_alone / 1e6:.2f} ms):")
for n, v in sorted(alone.items(), key=lambda kv: -kv[1])[:25]:
    print(f"  {v / 1e6:8.3f} ms  {n[:110]}")
