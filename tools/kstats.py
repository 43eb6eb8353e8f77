#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run: python tools/kstats.py <dir> [steps] -> ms/step per kernel."""
import csv
import glob
import sys

d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
import os
f = max(glob.glob(d + "/*/*kernel_stats.csv"), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot / 1e6 / steps:.2f} ms/step over {steps:g} steps")
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 45]:
    print(f"{float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms/step  x{int(r['Calls']) / steps:7.1f}  "
          f"avg {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:100]}")
