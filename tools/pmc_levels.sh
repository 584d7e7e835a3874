#!/bin/bash
# PMC comparison of the raster kernels of feature level 0 / 1 / 2 on the bench frame.
export TMPDIR=/tmp
for l in 0 1; do
  export RXR_MIN_KERNEL_LEVEL=$l
  rm -rf gpurun_out/lvlpmc$l
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d gpurun_out/lvlpmc$l -- python3 bench.py --steps 20 --warmup 2 --no-cpu > gpurun_out/lvlpmc$l.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/lvlpmcb$l -- python3 bench.py --steps 20 --warmup 2 --no-cpu > gpurun_out/lvlpmcb$l.log 2>&1
done
python3 - <<'PY'
import csv,glob,collections
for l in (0,1):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for d in (f"gpurun_out/lvlpmc{l}", f"gpurun_out/lvlpmcb{l}"):
        for f in glob.glob(d+"/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items():
        if k.startswith("k_raster"): print("level", l, k, {c: round(sum(x)/len(x)/1e6,2) for c,x in sorted(v.items())})
PY
