#!/bin/bash
# PMC counters of the raster kernel on one configuration of tools/run_configs.py.   usage: tools/pmc_config.sh C5s_shader [tag]
export TMPDIR=/tmp
CFG=$1; TAG=${2:-pmc}
rm -rf gpurun_out/${TAG}a gpurun_out/${TAG}b gpurun_out/${TAG}c
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d gpurun_out/${TAG}a -- python3 tools/run_configs.py --configs $CFG --oracle none --frames 6 > gpurun_out/${TAG}a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/${TAG}b -- python3 tools/run_configs.py --configs $CFG --oracle none --frames 6 > gpurun_out/${TAG}b.log 2>&1
rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_IFETCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM --output-format csv -d gpurun_out/${TAG}c -- python3 tools/run_configs.py --configs $CFG --oracle none --frames 6 > gpurun_out/${TAG}c.log 2>&1
python3 - "$TAG" <<'PY'
import csv,glob,collections,sys
tag=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for d in (f"gpurun_out/{tag}a", f"gpurun_out/{tag}b", f"gpurun_out/{tag}c"):
    for f in glob.glob(d+"/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if k.startswith("k_raster"): print(k, {c: round(sum(x)/len(x)/1e6,3) for c,x in sorted(v.items())})
PY
