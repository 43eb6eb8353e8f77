#!/usr/bin/env python3
"""Per-kernel SQ counter summary of a rocprofv3 --pmc run: python tools/pmcstats.py <dir> [name filter]."""
import collections
import csv
import glob
import os
import sys

f = max(glob.glob(sys.argv[1] + "/*/*counter_collection.csv"), key=os.path.getmtime)
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if flt not in r["Kernel_Name"]:
        continue
    k = r["Kernel_Name"][:44] + " g=" + r["Grid_Size"]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        cnt[k] += 1
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    n = cnt[k]
    w = v["SQ_WAVE_CYCLES"]
    d = sorted(dur[k])[n // 2]
    mf = v["SQ_VALU_MFMA_BUSY_CYCLES"] / n / 1024 / 2100  # us of MFMA issue per SIMD at 2.1 GHz
    conf = v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1)
    print(f"{k:66s} n={n:3d} med {d:7.1f} us  mfma_busy {mf / d:.2f}  wait_any {v['SQ_WAIT_ANY'] / w:.2f}  "
          f"wait_inst {v['SQ_WAIT_INST_ANY'] / w:.2f}  active {v['SQ_ACTIVE_INST_ANY'] / w:.2f}  lds_conflict/active {conf:.2f}")
