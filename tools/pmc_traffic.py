#!/usr/bin/env python3
"""Aggregate two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs) into profiles/r01_pmc_hbm_traffic.json:
python tools/pmc_traffic.py <fetch_dir> <write_dir> <steps_profiled> [out.json]"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_hash  # noqa: E402  (a profile is quoted by bench.py only while the sources it came from are current)


def load(d):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        a = agg[r["Kernel_Name"].split("(")[0][:60]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return agg


F, W = load(sys.argv[1]), load(sys.argv[2])
steps = float(sys.argv[3])
out = sys.argv[4] if len(sys.argv) > 4 else "profiles/r02_pmc_hbm_traffic.json"
rows, tf, tw = [], 0.0, 0.0
for k in F:
    n = F[k][0]
    fb = F[k][1] * 1024 * 2  # KB -> B; x2: gfx950 tallies 128-B requests at 64 B (MI355X_MICROARCH.md, HBM)
    wb = W.get(k, [0, 0.0])[1] * 1024
    tf += fb
    tw += wb
    rows.append((fb + wb, k, n, fb, wb))
rows.sort(reverse=True)
print(f"per step: fetch {tf / steps / 1e9:.1f} GB, write {tw / steps / 1e9:.1f} GB")
for t, k, n, fb, wb in rows[:14]:
    print(f"{k:62s} x{n / steps:6.1f}/step  fetch {fb / n / 1e6:8.1f} MB  write {wb / n / 1e6:8.1f} MB per launch  {t / steps / 1e9:6.2f} GB/step")
# the GPU box has no .git: the caller stamps the commit the snapshot was taken from (tools/final_measure.sh is started as
# CTU_COMMIT=$(git rev-parse --short HEAD) gpurun ... with the variable expanded in the build container)
commit = os.environ.get("CTU_COMMIT", "")
if not commit:
    try:
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "?"
    except OSError:
        commit = "?"
json.dump({"source_hash": source_hash(), "commit": commit, "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over python bench.py --serial --steps 2 "
                   "--warmup 1 (single stream: a launch's counters are that kernel's alone); FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md section HBM); bytes per launch "
                   "averaged over the launches of a step",
           "per_step_GB": {"fetch": tf / steps / 1e9, "write": tw / steps / 1e9},
           "kernels": {k: {"launches_per_step": n / steps, "fetch_bytes_per_launch": fb / n, "write_bytes_per_launch": wb / n}
                       for t, k, n, fb, wb in rows}}, open(out, "w"), indent=1)
