#!/usr/bin/env python3
"""Per-launch durations of the kernels whose name contains one of the given substrings, from a rocprofv3 --kernel-trace csv
directory: python tools/trace_kernels.py DIR substr [substr ...]"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
d = collections.defaultdict(list)
tot = collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot[n[:70]] += us
    if any(s in n for s in sys.argv[2:]):
        d[n[:70]].append(us)
for n, v in d.items():
    v2 = sorted(v)
    print(n, len(v), "min %.0f med %.0f max %.0f us" % (v2[0], v2[len(v2) // 2], v2[-1]))
    print("   ", " ".join("%.0f" % x for x in v[:48]))
print("top kernels by total time (ms):")
for n, t in sorted(tot.items(), key=lambda kv: -kv[1])[:12]:
    print("  %9.2f  %s" % (t / 1e3, n))
