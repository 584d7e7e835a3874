#!/bin/bash
# rocprofv3 passes over one configuration of tools/run_configs.py: kernel trace + separate PMC passes (SQ x2, FETCH_SIZE,
# WRITE_SIZE; counters never combined with traces, the program directly after `--`); the summaries land in
# gpurun_out/<tag>/{kernel_stats.csv,pmc_summary.json}.   usage: tools/profile_config.sh <tag> <config> [run_configs args...]
set -u
TAG=$1; CFG=$2; shift 2
export TMPDIR=/tmp
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
ARGS="tools/run_configs.py --configs $CFG --oracle none --frames 8 --no-e2e $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 $ARGS > "$OUT/kt.log" 2>&1
echo "kt done"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS \
    --output-format csv -d "$OUT/pmc_sq1" -- python3 $ARGS > "$OUT/pmc_sq1.log" 2>&1
echo "sq1 done"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32 SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM \
    --output-format csv -d "$OUT/pmc_sq2" -- python3 $ARGS > "$OUT/pmc_sq2.log" 2>&1
echo "sq2 done"
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_fetch" -- python3 $ARGS > "$OUT/pmc_fetch.log" 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 $ARGS > "$OUT/pmc_write.log" 2>&1
echo "write done"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum --output-format csv -d "$OUT/pmc_tcc" -- python3 $ARGS > "$OUT/pmc_tcc.log" 2>&1
echo "tcc done"
grep -l "Memory access fault" "$OUT"/*.log && echo "FAULT DETECTED"
python3 tools/summarize_prof.py "$OUT" | grep -v "at::native\|rocclr"
# the raw per-dispatch CSVs are large; keep the summaries only
rm -rf "$OUT"/kt "$OUT"/pmc_sq1 "$OUT"/pmc_sq2 "$OUT"/pmc_fetch "$OUT"/pmc_write "$OUT"/pmc_tcc
