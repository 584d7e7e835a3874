#!/bin/bash
# A/B of k_raster_vm builds (build/variants/librxr_hip_*.so) on the shader cost probe; scratch tool
cd "$(dirname "$0")/.."
cp rusterix_amd/csrc/librxr_hip.so /tmp/librxr_hip_orig.so
trap 'cp /tmp/librxr_hip_orig.so rusterix_amd/csrc/librxr_hip.so' EXIT   # the product library comes back on ANY exit
for so in build/variants/librxr_hip_*.so; do
  name=$(basename "$so" .so); name=${name#librxr_hip_}
  cp "$so" rusterix_amd/csrc/librxr_hip.so
  echo "== $name"; timeout 300 python tools/vm_cost.py 2>&1 | tail -6
done
cp /tmp/librxr_hip_orig.so rusterix_amd/csrc/librxr_hip.so
