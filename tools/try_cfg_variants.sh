#!/bin/bash
# Swaps kernel-variant builds (build/variants/librxr_hip_<name>.so) in on the GPU box and runs tools/run_configs.py
# for each.  Scratch tool for tuning experiments.   usage: tools/try_cfg_variants.sh C5s,C5 [names...]
set -u
cd "$(dirname "$0")/.."
CFG=$1; shift
cp rusterix_amd/csrc/librxr_hip.so /tmp/librxr_hip_orig.so
trap 'cp /tmp/librxr_hip_orig.so rusterix_amd/csrc/librxr_hip.so' EXIT   # the product library comes back on ANY exit
for name in base "$@"; do
  if [ "$name" != base ]; then cp "build/variants/librxr_hip_$name.so" rusterix_amd/csrc/librxr_hip.so; fi
  timeout 300 python tools/run_configs.py --configs "$CFG" --oracle none --frames 20 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$name', d['config'], 'setup_us', d['setup_kernels_us'], 'raster_us', d['raster_kernel_us'], 'frame_ms', d['frame_ms_device_resident'])
    elif 'fault' in l.lower() or 'error' in l.lower(): print('$name', l.strip())
"
done
cp /tmp/librxr_hip_orig.so rusterix_amd/csrc/librxr_hip.so
