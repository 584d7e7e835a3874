#!/bin/bash
# Swaps kernel-variant builds of librxr_hip.so (build/variants/librxr_hip_<name>.so) in on the GPU box and
# runs a short bench for each.  Scratch tool for tuning experiments; results go to gpurun_out/.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
cp rusterix_amd/csrc/librxr_hip.so /tmp/librxr_hip_orig.so
trap 'cp /tmp/librxr_hip_orig.so rusterix_amd/csrc/librxr_hip.so' EXIT   # the product library comes back on ANY exit
for so in build/variants/librxr_hip_*.so; do
  name=$(basename "$so" .so); name=${name#librxr_hip_}
  cp "$so" rusterix_amd/csrc/librxr_hip.so
  for L in "$@"; do
    out=$(timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu --lights "$L" 2>&1 | tail -1)
    if echo "$out" | grep -q "Memory access fault"; then echo "GPU_FAULT in $name"; cp /tmp/librxr_hip_orig.so rusterix_amd/csrc/librxr_hip.so; exit 1; fi
    echo "$out" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', 'lights', $L, 'Mpx/s', d['value'], 'ms', d['ms_per_step'], 'raster_us', d['roofline']['kernel_avg_us'])"
  done
done
cp /tmp/librxr_hip_orig.so rusterix_amd/csrc/librxr_hip.so
