#!/bin/bash
# rocprofv3 passes over an arbitrary python command (kernel trace + separate PMC passes, never combined);
# summaries land in gpurun_out/<tag>/.   usage: tools/profile_cmd.sh <tag> <script.py> [args...]
set -u
TAG=$1; shift
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$@" > "$OUT/kt.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS \
    --output-format csv -d "$OUT/pmc_sq1" -- python3 "$@" > "$OUT/pmc_sq1.log" 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM \
    --output-format csv -d "$OUT/pmc_sq2" -- python3 "$@" > "$OUT/pmc_sq2.log" 2>&1
grep -l "Memory access fault" "$OUT"/*.log && echo "FAULT DETECTED"
python3 tools/summarize_prof.py "$OUT" | grep -v "at::native\|rocclr"
