#!/usr/bin/env python3
"""Gaps between consecutive dispatches of a rocprofv3 --kernel-trace CSV (Start_Timestamp / End_Timestamp in ns): how much of a step is
kernels and how much is the space between them.   usage: tools/trace_gaps.py <dir with *_kernel_trace.csv> [kernel name substrings...]"""
import csv, glob, os, sys
import numpy as np

d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].split("(")[0] for r in rows]
st = np.array([int(r["Start_Timestamp"]) for r in rows], dtype=np.int64)
en = np.array([int(r["End_Timestamp"]) for r in rows], dtype=np.int64)
keep = sys.argv[2:] or ["k_setup3d", "k_raster"]
idx = [i for i, n in enumerate(names) if any(k in n for k in keep)]
pairs = {}
for a, b in zip(idx[:-1], idx[1:]):
    if b != a + 1:
        continue  # something else ran in between
    pairs.setdefault((names[a], names[b]), []).append((st[b] - en[a]) / 1e3)
for (a, b), g in sorted(pairs.items()):
    g = np.array(g)
    print(f"{a:>18s} -> {b:<18s} n={len(g):5d}  gap us: median {np.median(g):7.2f}  p10 {np.percentile(g, 10):7.2f}  p90 {np.percentile(g, 90):7.2f}")
dur = {}
for i in idx:
    dur.setdefault(names[i], []).append((en[i] - st[i]) / 1e3)
for n, v in sorted(dur.items()):
    print(f"{n:>18s} n={len(v):5d} duration us: median {np.median(v):8.2f} mean {np.mean(v):8.2f}")
