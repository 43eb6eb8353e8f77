#!/usr/bin/env python3
"""Per hardware queue of ONE step of a rocprofv3 --kernel-trace run: kernels, busy time, and for the queues with few launches the
list of their kernels with start times - which stream landed on which queue, and what each queue does while another waits.
python tools/queue_picture.py <kernel_trace.csv> [steps from the end]"""
import csv
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]))
rows.sort()
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
marks = [r[0] for r in rows if "im2col_cin1" in r[3]][::2]
t0, t1 = marks[-back - 1], marks[-back]
rows = [r for r in rows if t0 <= r[0] < t1]
print(f"step window {1e-6 * (t1 - t0):.2f} ms, {len(rows)} kernels")
byq = defaultdict(list)
for r in rows:
    byq[r[2]].append(r)
for q in sorted(byq, key=lambda q: -len(byq[q])):
    iv = byq[q]
    busy = sum(e - s for s, e, _, _ in iv)
    names = defaultdict(int)
    for _, _, _, n in iv:
        names[n.split("(")[0][:40]] += 1
    top = ", ".join(f"{k} x{v}" for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:4])
    print(f"queue {q:>3s}: {len(iv):5d} kernels, busy {1e-6 * busy:7.2f} ms, active {1e-6 * (iv[0][0] - t0):6.2f} .. {1e-6 * (max(e for _, e, _, _ in iv) - t0):6.2f} ms   [{top}]")
# device-level: time with 0 / 1 / 2 / 3+ kernels in flight
ev = sorted([(s, 1) for s, _, _, _ in rows] + [(e, -1) for _, e, _, _ in rows])
hist, k, last = defaultdict(int), 0, t0
for t, d in ev:
    hist[min(k, 3)] += t - last
    last = t
    k += d
print("kernels in flight: " + "  ".join(f"{n}{'+' if n == 3 else ''}: {1e-6 * hist[n]:.2f} ms" for n in range(4)))
