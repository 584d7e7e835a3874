#!/bin/bash
# VALU instruction counts by class per launch of every kernel of one tools/run_configs.py configuration (two rocprofv3 PMC
# passes).   usage: tools/valu_counts_cfg.sh <tag> <config> [run_configs args...]
set -u
TAG=$1; CFG=$2; shift 2
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
ARGS="tools/run_configs.py --configs $CFG --oracle none --frames 6 $*"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT \
    --output-format csv -d "$OUT/pmc_a" -- python3 $ARGS > "$OUT/a.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES \
    --output-format csv -d "$OUT/pmc_b" -- python3 $ARGS > "$OUT/b.log" 2>&1
python3 - "$OUT" "$CFG $*" <<'PY'
import csv, glob, sys, collections, json
out, args = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc_*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    if not k.startswith("k_"): continue
    m = {c: sum(x) / len(x) for c, x in v.items()}
    waves = max(1.0, m.get("SQ_WAVES", 1))
    cls = {c.replace("SQ_INSTS_VALU_", "").lower(): m.get(c, 0) for c in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32", "SQ_INSTS_VALU_CVT", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64")}
    other = m.get("SQ_INSTS_VALU", 0) - sum(cls.values())
    print(json.dumps(dict(args=args, kernel=k, waves=int(waves), valu_M=round(m.get("SQ_INSTS_VALU", 0) / 1e6, 2), valu_per_wave=round(m.get("SQ_INSTS_VALU", 0) / waves, 1),
               salu_per_wave=round(m.get("SQ_INSTS_SALU", 0) / waves, 1), per_wave={c: round(x / waves, 1) for c, x in cls.items()}, other_per_wave=round(other / waves, 1),
               lane_util=round(m.get("SQ_THREAD_CYCLES_VALU", 0) / max(1.0, m.get("SQ_ACTIVE_INST_VALU", 1) * 64), 3),
               lds_per_wave=round(m.get("SQ_INSTS_LDS", 0) / waves, 1), vmem_per_wave=round(m.get("SQ_INSTS_VMEM", 0) / waves, 1), smem_per_wave=round(m.get("SQ_INSTS_SMEM", 0) / waves, 1))))
PY
rm -rf "$OUT"/pmc_a "$OUT"/pmc_b
