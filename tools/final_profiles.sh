#!/bin/bash
# End-of-round measurement set (run on the GPU box): the bench line, every configuration host- and device-projected with parity
# against the oracle (full size, C5 and C5 + program included), end-to-end on multi-member contexts.   usage: tools/final_profiles.sh <tag>
set -u
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
python3 bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_line.err"
echo "bench line done"
python3 tools/run_configs.py --configs C2,C3,C4,C5s,C5s_shader,D2 --oracle C2,C3,C4,C5s,C5s_shader,D2 --frames 30 > "$OUT/configs.jsonl" 2> "$OUT/configs.err"
python3 tools/run_configs.py --configs C5,C5shader --oracle C5,C5shader --frames 30 >> "$OUT/configs.jsonl" 2>> "$OUT/configs.err"
python3 tools/run_configs.py --configs C5s_shader,C5shader --oracle C5s_shader,C5shader --frames 30 --jit 1 > "$OUT/configs_jit.jsonl" 2> "$OUT/configs_jit.err"
echo "configs done"
python3 tools/run_configs.py --configs C2,C3,C4,C5s,C5s_shader --oracle C2,C3,C4,C5s,C5s_shader --frames 30 --device-projection > "$OUT/configs_devproj.jsonl" 2> "$OUT/configs_devproj.err"
python3 tools/run_configs.py --configs C5,C5shader --oracle C5 --frames 30 --device-projection >> "$OUT/configs_devproj.jsonl" 2>> "$OUT/configs_devproj.err"
echo "device projection configs done"
python3 tools/e2e_multi.py --members 1,2,4,8 > "$OUT/e2e_multi.jsonl" 2> "$OUT/e2e_multi.err"
python3 tools/e2e_multi.py --members 1,4 --config C5 --device-projection >> "$OUT/e2e_multi.jsonl" 2>> "$OUT/e2e_multi.err"
echo "multi-member end-to-end done"
python3 tools/show_cfg.py "$OUT/configs.jsonl" "$OUT/configs_devproj.jsonl"
