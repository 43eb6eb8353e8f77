#!/usr/bin/env python3
"""LDS / MFMA occupancy of the halo kernels from one rocprofv3 --pmc run:
python tools/pmc_lds.py <dir>   (counters SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES)
Prints, per kernel shape (name + grid): launches, median duration, and each counter per launch together with its ratio to
duration x 256 CUs x clock (LDS counters tick once per LDS-array cycle per CU; MFMA busy once per cycle per SIMD)."""
import collections, csv, glob, os, sys
f = max(glob.glob(sys.argv[1] + "/*/*counter_collection.csv"), key=os.path.getmtime)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "halo" not in r["Kernel_Name"]:
        continue
    k = r["Kernel_Name"][:40] + " g=" + r["Grid_Size"]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    n = len(dur[k])
    d = sorted(dur[k])[n // 2]
    line = f"{k:60s} n={n:3d} med {d:7.1f} us"
    for c in ("SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES"):
        if c in v:
            line += f"  {c[3:]} {v[c] / n:.3e}"
    print(line)
