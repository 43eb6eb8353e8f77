#!/usr/bin/env python3
"""MFMA-pipe and LDS occupancy per kernel family of one rocprofv3 --pmc run over `python bench.py --serial`:
python tools/pmc_mfma.py <dir> <steps profiled>   (counters SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES)
Per kernel name: launches per step, summed duration per step, MFMA-busy share = busy cycles / (1 024 SIMDs x duration x 1.9 GHz - the
clock the chip holds under matrix load, profiles/r02_peaks.log), LDS-active share = LDS_IDX_ACTIVE / (256 CUs x duration x 1.9 GHz),
bank conflicts per active LDS cycle."""
import collections
import csv
import glob
import os
import sys

f = max(glob.glob(sys.argv[1] + "/*/*counter_collection.csv"), key=os.path.getmtime)
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
GHZ = 1.9
agg = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float)
cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:64]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        cnt[k] += 1
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3   # us
tot = sum(dur.values()) / steps
print(f"# {tot / 1e3:.2f} ms of kernel time per step (single stream, counters on); MFMA share of the whole step: "
      f"{sum(v['SQ_VALU_MFMA_BUSY_CYCLES'] for v in agg.values()) / (1024 * sum(dur.values()) * GHZ * 1e3):.3f}")
print(f"{'kernel':66s} {'x/step':>7s} {'ms/step':>8s} {'mfma busy':>9s} {'lds active':>10s} {'conflicts':>9s}")
for k in sorted(dur, key=lambda k: -dur[k]):
    if dur[k] / steps < 50.0:
        continue
    v = agg[k]
    cyc = dur[k] * GHZ * 1e3
    print(f"{k:66s} {cnt[k] / steps:7.1f} {dur[k] / steps / 1e3:8.3f} {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):9.3f} "
          f"{v['SQ_LDS_IDX_ACTIVE'] / (256 * cyc):10.3f} {v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_LDS_IDX_ACTIVE'], 1.0):9.3f}")
