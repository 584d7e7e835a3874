#!/bin/bash
# Profiles `python3 bench.py` on the GPU box: one --kernel-trace --stats pass and separate --pmc passes
# (counters never combined with traces), then writes the summaries a reviewer needs into
# gpurun_out/<tag>/ ; copy them to profiles/ afterwards.   usage: tools/profile_bench.sh <tag> [bench args]
set -u
TAG=${1:-prof}; shift || true
EXTRA="$*"
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
KT="--steps 100 --warmup 10 --no-cpu --no-e2e --no-in-flight $EXTRA"
PM="--steps 20 --warmup 2 --no-cpu --no-e2e --no-in-flight $EXTRA"
export RXR_BENCH_MIN_TIMED_S=0.05   # the counter passes need launches, not a long timed region
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 bench.py $KT > "$OUT/kt.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS \
    --output-format csv -d "$OUT/pmc_sq1" -- python3 bench.py $PM > "$OUT/pmc_sq1.log" 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32 SQ_LDS_BANK_CONFLICT \
    --output-format csv -d "$OUT/pmc_sq2" -- python3 bench.py $PM > "$OUT/pmc_sq2.log" 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py $PM > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 bench.py $PM > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_sq3" -- python3 bench.py $PM > "$OUT/pmc_sq3.log" 2>&1
grep -l "Memory access fault" "$OUT"/*.log && echo "FAULT DETECTED"
python3 tools/summarize_prof.py "$OUT"
rm -rf "$OUT"/kt "$OUT"/pmc_sq1 "$OUT"/pmc_sq2 "$OUT"/pmc_sq3 "$OUT"/pmc_fetch "$OUT"/pmc_write
