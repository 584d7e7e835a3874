#!/usr/bin/env python3
"""Condenses the rocprofv3 output of tools/profile_bench.sh into two small files:
<dir>/kernel_stats.csv (copy of the --stats summary) and <dir>/pmc_summary.json (per kernel: mean counter
value per launch; FETCH_SIZE / WRITE_SIZE are KiB in rocprofv3's output and are converted to bytes, with
the gfx950 x2 correction for FETCH_SIZE given separately)."""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

out = sys.argv[1]
ks = glob.glob(os.path.join(out, "kt", "**", "*_kernel_stats.csv"), recursive=True)
if ks:
    shutil.copy(ks[0], os.path.join(out, "kernel_stats.csv"))
    print(open(ks[0]).read()[:1500])
bench = [l for l in open(os.path.join(out, "kt.log")) if l.startswith("{")] if os.path.exists(os.path.join(out, "kt.log")) else []
summary = defaultdict(dict)
for d in sorted(glob.glob(os.path.join(out, "pmc*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                summary[k][c] = sum(v) / len(v)
                summary[k]["launches_" + os.path.basename(d)] = len(v)
for k, cs in summary.items():
    if "FETCH_SIZE" in cs:
        cs["fetch_bytes_raw"] = cs["FETCH_SIZE"] * 1024
        cs["fetch_bytes_x2_gfx950"] = cs["FETCH_SIZE"] * 2048
    if "WRITE_SIZE" in cs:
        cs["write_bytes"] = cs["WRITE_SIZE"] * 1024
    if "SQ_THREAD_CYCLES_VALU" in cs and cs.get("SQ_ACTIVE_INST_VALU"):
        cs["valu_lane_utilisation"] = cs["SQ_THREAD_CYCLES_VALU"] / (cs["SQ_ACTIVE_INST_VALU"] * 64.0)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import bench as _bench  # the hash of the kernel sources these counters were taken on (bench.py quotes it with the numbers)
    sha = _bench.kernel_sources_sha()
except Exception:
    sha = None
res = {"bench_line": json.loads(bench[-1]) if bench else None, "kernel_sources_sha": sha, "kernels": summary}
json.dump(res, open(os.path.join(out, "pmc_summary.json"), "w"), indent=1)
for k, cs in summary.items():
    print(k, {c: round(v, 1) for c, v in cs.items() if not c.startswith("launches")})
