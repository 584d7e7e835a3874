#!/bin/bash
# runs tools/phase_timing.py with the phase-timing variant (build/variants/librxr_hip_phase.so: tools/build_variant.sh phase -DRXR_PHASE_TIMING=1)
cd "$(dirname "$0")/.."
cp rusterix_amd/csrc/librxr_hip.so /tmp/librxr_hip_orig.so
trap 'cp /tmp/librxr_hip_orig.so rusterix_amd/csrc/librxr_hip.so' EXIT
cp build/variants/librxr_hip_phase.so rusterix_amd/csrc/librxr_hip.so
for args in "$@"; do python3 tools/phase_timing.py $args; done
