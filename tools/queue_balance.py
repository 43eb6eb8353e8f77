#!/usr/bin/env python3
"""Per hardware queue of a rocprofv3 --kernel-trace run: busy time (union of kernel intervals), first start and last end, over the
last `frac` of the trace - is one of the two branch streams of CTUNet idle while the other finishes?
python tools/queue_balance.py <kernel_trace.csv> [frac]"""
import csv
import sys
from collections import defaultdict

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]))
rows.sort()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
t0, t1 = rows[0][0], max(r[1] for r in rows)
cut = t1 - int((t1 - t0) * frac)
rows = [r for r in rows if r[0] >= cut]
# step boundaries: the AdamW kernel(s) end a step
marks = [r[1] for r in rows if "adamw_kernel" in r[3]]
print(f"window {1e-6 * (t1 - cut):.1f} ms, {len(rows)} kernels, optimizer kernels seen: {len(marks)}")
byq = defaultdict(list)
for s, e, q, n in rows:
    byq[q].append((s, e))
for q, iv in sorted(byq.items()):
    iv.sort()
    busy, cs, ce = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    print(f"queue {q}: {len(iv):6d} kernels, busy {1e-6 * busy:8.2f} ms = {100.0 * busy / (t1 - cut):5.1f} % of the window")
# gaps in the union over all queues
allv = sorted((s, e) for s, e, _, _ in rows)
idle, ce = 0, allv[0][1]
for s, e in allv[1:]:
    if s > ce:
        idle += s - ce
    ce = max(ce, e)
print(f"device idle (no kernel on any queue): {1e-6 * idle:.2f} ms = {100.0 * idle / (t1 - cut):.1f} %")
