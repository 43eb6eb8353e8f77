#!/usr/bin/env python3
"""Duration histogram per kernel name from a rocprofv3 --kernel-trace CSV: python tools/durhist.py <kernel_trace.csv> <substring> [steps]
-> launches and summed time per duration bucket (us), so that launch-floor-bound and bandwidth-bound launches of one kernel separate."""
import csv
import sys

path, sub = sys.argv[1], sys.argv[2]
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
edges = [0, 6, 8, 10, 12, 15, 20, 25, 30, 40, 60, 100, 200, 400, 1e9]
cnt = [0] * (len(edges) - 1)
tot = [0.0] * (len(edges) - 1)
for r in csv.DictReader(open(path)):
    if sub not in r["Kernel_Name"]:
        continue
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for i in range(len(edges) - 1):
        if edges[i] <= us < edges[i + 1]:
            cnt[i] += 1
            tot[i] += us
            break
print(f"{sub}: {sum(cnt) / steps:.1f} launches/step, {sum(tot) / 1e3 / steps:.3f} ms/step")
for i in range(len(edges) - 1):
    if cnt[i]:
        hi = "inf" if edges[i + 1] > 1e8 else f"{edges[i + 1]:g}"
        print(f"  {edges[i]:>4g} - {hi:>4s} us: x{cnt[i] / steps:7.1f}/step  {tot[i] / 1e3 / steps:7.3f} ms/step")
