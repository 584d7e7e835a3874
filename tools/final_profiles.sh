#!/bin/bash
# End-of-round measurement set (run on the GPU box): bench profile + bench line, every configuration host- and
# device-projected with parity, the C5 program configuration under the profiler.   usage: tools/final_profiles.sh <tag>
set -u
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
bash tools/profile_bench.sh $TAG/bench > "$OUT/bench_profile.log" 2>&1
echo "bench profile done"
python3 bench.py > "$OUT/bench_line.json" 2> "$OUT/bench_line.err"
echo "bench line done"
python3 tools/run_configs.py --configs C2,C3,C4,C5s,C5s_shader,D2 --oracle C2,C3,C4,C5s,C5s_shader,D2 --frames 30 > "$OUT/configs.jsonl" 2> "$OUT/configs.err"
python3 tools/run_configs.py --configs C5,C5shader --oracle none --frames 30 >> "$OUT/configs.jsonl" 2>> "$OUT/configs.err"
echo "configs done"
python3 tools/run_configs.py --configs C2,C3,C4,C5s,C5s_shader --oracle C2,C3,C4,C5s,C5s_shader --frames 30 --device-projection > "$OUT/configs_devproj.jsonl" 2> "$OUT/configs_devproj.err"
python3 tools/run_configs.py --configs C5,C5shader --oracle none --frames 30 --device-projection >> "$OUT/configs_devproj.jsonl" 2>> "$OUT/configs_devproj.err"
echo "device projection configs done"
bash tools/profile_cmd.sh $TAG/c5shader tools/run_configs.py --configs C5shader --oracle none --frames 10 > "$OUT/c5shader_profile.log" 2>&1
echo "c5shader profile done"
python3 tools/run_configs.py --configs C5shader --oracle C5shader --frames 5 > "$OUT/c5shader_full_parity.jsonl" 2> "$OUT/c5shader_full_parity.err"
echo "c5shader full-size parity done"
