#!/bin/bash
# SQ counters of C5's k_raster_rows for the in-tree library and the named variants (timing / counter diagnostics; variants may render wrong frames)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
cp rusterix_amd/csrc/librxr_hip.so /tmp/new.so
trap 'cp /tmp/new.so rusterix_amd/csrc/librxr_hip.so' EXIT
for name in new "$@"; do
  if [ $name = new ]; then cp /tmp/new.so rusterix_amd/csrc/librxr_hip.so; else cp build/variants/librxr_hip_$name.so rusterix_amd/csrc/librxr_hip.so; fi
  OUT=gpurun_out/pmcv_$name; mkdir -p $OUT
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/pmc -- python3 tools/run_configs.py --configs C5 --oracle none --frames 6 --no-e2e > $OUT/log.txt 2>&1
  python3 - "$OUT" "$name" <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob(sys.argv[1] + "/pmc/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("k_raster_rows"):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: round(sum(v) / len(v) / 1e6, 2) for k, v in sorted(acc.items())}, "(millions per launch)")
PY
  rm -rf $OUT/pmc
done
