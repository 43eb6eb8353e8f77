#!/usr/bin/env python3
"""Who waits for whom in the overlapped step?  python tools/critpath.py <kernel_trace.csv> [step index from the end, default 2]
Takes ONE whole step of a rocprofv3 --kernel-trace run of bench.py (steps are delimited by the im2col_cin1 launch that opens a
forward pass) and prints, per hardware queue, its busy time, its idle gaps longer than 150 us (with the kernels on either side and
what the other queues ran meanwhile) and a 1-ms raster of the step: per queue the kernel family that held it longest."""
import csv
import sys
from collections import defaultdict


def fam(n):
    for k, v in (("conv3_halo_wgrad", "Hw"), ("conv3_halo_dma", "Hf"), ("halo_", "hr"), ("in_bwd", "Ib"), ("in_apply", "If"), ("in_", "I."),
                 ("gemm_nt_stream", "Gs"), ("gemm_nt", "Gn"), ("gemm_tn", "Gt"), ("tn_reduce", "gr"), ("igemm", "Gg"), ("attn", "At"),
                 ("layernorm", "Ln"), ("pwa", "Pw"), ("adamw", "Op"), ("dicece", "Lo"), ("pack_frag", "pk"), ("pixel_shuffle", "Ps")):
        if k in n:
            return v
    return ".."


rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]))
rows.sort()
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
marks = [r[0] for r in rows if "im2col_cin1" in r[3]]
marks = marks[::2]   # two launches per forward pass (stem + vit_encoder0's first conv) -> one mark per step
t0, t1 = marks[-back - 1], marks[-back]
rows = [r for r in rows if t0 <= r[0] < t1]
print(f"step window {1e-6 * (t1 - t0):.2f} ms, {len(rows)} kernels")
byq = defaultdict(list)
for r in rows:
    byq[r[2]].append(r)
qs = sorted(byq, key=lambda q: -sum(e - s for s, e, _, _ in byq[q]))
for q in qs:
    iv = byq[q]
    busy = sum(e - s for s, e, _, _ in iv)
    print(f"queue {q}: {len(iv):5d} kernels, busy {1e-6 * busy:7.2f} ms, first {1e-6 * (iv[0][0] - t0):6.2f} ms, last end {1e-6 * (max(e for _, e, _, _ in iv) - t0):6.2f} ms")
print("\nidle gaps > 150 us per queue (queue, at ms, length us, kernel before -> kernel after | families busy elsewhere meanwhile)")
for q in qs[:4]:
    iv = byq[q]
    for a, b in zip(iv, iv[1:]):
        gap = b[0] - a[1]
        if gap > 150000:
            others = defaultdict(int)
            for s, e, qq, n in rows:
                if qq != q and e > a[1] and s < b[0]:
                    others[fam(n)] += min(e, b[0]) - max(s, a[1])
            oth = " ".join(f"{k}:{v / 1e3:.0f}" for k, v in sorted(others.items(), key=lambda kv: -kv[1])[:5])
            print(f"  q{q} @{1e-6 * (a[1] - t0):6.2f} ms  {gap / 1e3:7.0f} us  {a[3][:38]:38s} -> {b[3][:38]:38s} | {oth}")
print("\nraster: per ms and queue, family that ran longest (busy fraction 0-9); families: Hf halo fwd/dgrad, Hw halo wgrad, I* InstanceNorm, G* GEMMs, At attention, Ln LayerNorm")
nms = int((t1 - t0) / 1e6) + 1
for q in qs[:5]:
    line = []
    for m in range(nms):
        a, b = t0 + m * 1000000, t0 + (m + 1) * 1000000
        acc = defaultdict(int)
        for s, e, _, n in byq[q]:
            if e > a and s < b:
                acc[fam(n)] += min(e, b) - max(s, a)
        if acc:
            k = max(acc, key=acc.get)
            line.append(f"{k}{min(9, int(10 * sum(acc.values()) / 1e6))}")
        else:
            line.append(" . ")
    print(f"q{q:>3s} " + " ".join(line))
