#!/bin/bash
# A-B-A-B comparison of a kernel-variant build against the default on one box.  usage: tools/abab.sh <variant> <lights...>
cd "$(dirname "$0")/.."
V=$1; shift
cp rusterix_amd/csrc/librxr_hip.so /tmp/orig.so
trap 'cp /tmp/orig.so rusterix_amd/csrc/librxr_hip.so' EXIT   # the product library comes back on ANY exit
for round in 1 2; do
  for which in base $V; do
    if [ $which = base ]; then cp /tmp/orig.so rusterix_amd/csrc/librxr_hip.so; else cp build/variants/librxr_hip_$which.so rusterix_amd/csrc/librxr_hip.so; fi
    for L in "$@"; do
      python bench.py --steps 200 --warmup 20 --no-cpu --lights $L 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$which', 'lights', $L, 'ms', d['ms_per_step'], 'raster_us', d['roofline']['kernel_avg_us'])"
    done
  done
done
cp /tmp/orig.so rusterix_amd/csrc/librxr_hip.so
