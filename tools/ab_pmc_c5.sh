#!/bin/bash
# Counters and phase shares of C5 for the in-tree library and build/variants/librxr_hip_prev.so (+ the pt_new / pt_prev phase-timing
# builds when present), one box.   usage: tools/ab_pmc_c5.sh <tag>
cd "$(dirname "$0")/.."
TAG=${1:-ab}
cp rusterix_amd/csrc/librxr_hip.so /tmp/new.so
trap 'cp /tmp/new.so rusterix_amd/csrc/librxr_hip.so' EXIT
for name in new prev; do
  if [ $name = prev ]; then cp build/variants/librxr_hip_prev.so rusterix_amd/csrc/librxr_hip.so; else cp /tmp/new.so rusterix_amd/csrc/librxr_hip.so; fi
  tools/profile_config.sh ${TAG}_$name C5 > gpurun_out/${TAG}_$name.log 2>&1
  grep "k_raster_rows\|FAULT" gpurun_out/${TAG}_$name.log | cut -c1-1500
  if [ -f build/variants/librxr_hip_pt_$name.so ]; then
    cp build/variants/librxr_hip_pt_$name.so rusterix_amd/csrc/librxr_hip.so
    python tools/phase_timing.py --scene c5 > gpurun_out/${TAG}_phase_$name.txt 2>&1; cat gpurun_out/${TAG}_phase_$name.txt
  fi
done
