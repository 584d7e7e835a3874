#!/bin/bash
# A-B-A-B of the in-tree library ("new") against build/variants/librxr_hip_<name>.so for every name given: C5 (or $CFGS) through
# tools/run_configs.py, host-projected.   usage: [CFGS=C5,C2] tools/ab_variants.sh name...
cd "$(dirname "$0")/.."
cp rusterix_amd/csrc/librxr_hip.so /tmp/new.so
trap 'cp /tmp/new.so rusterix_amd/csrc/librxr_hip.so' EXIT   # the product library comes back on ANY exit
for r in 1 2; do
for name in new "$@"; do
  if [ $name = new ]; then cp /tmp/new.so rusterix_amd/csrc/librxr_hip.so; else cp build/variants/librxr_hip_$name.so rusterix_amd/csrc/librxr_hip.so; fi
  timeout 300 python tools/run_configs.py --configs ${CFGS:-C5} --oracle none --frames 20 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$name', d['config'], 'setup_us', d['setup_kernels_us'], 'raster_us', d['raster_kernel_us'], 'frame_ms', d['frame_ms_device_resident_no_events'])
"
done; done
