#!/bin/bash
# Interpreter cost probes (tools/run_configs.py C5s_vm:*) for the in-tree library and kernel-variant builds.   usage: tools/vm_probe.sh [names...]
cd "$(dirname "$0")/.."
cp rusterix_amd/csrc/librxr_hip.so /tmp/librxr_hip_orig.so
# the product library is put back on ANY exit (an interrupted run must not leave a variant build behind for the tests / bench)
trap 'cp /tmp/librxr_hip_orig.so rusterix_amd/csrc/librxr_hip.so' EXIT
for name in base "$@"; do
  if [ "$name" != base ]; then cp "build/variants/librxr_hip_$name.so" rusterix_amd/csrc/librxr_hip.so; fi
  for v in C5s_vm:empty C5s_vm:u40 C5s_vm:b20 C5s_vm:g20 C5s_vm:m20 C5s_shader C5shader; do
    timeout 300 python tools/run_configs.py --configs $v --oracle none --frames 20 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$name', d['config'], 'raster_us', d['raster_kernel_us'])
    elif 'rror' in l: print('$name', l.strip())
"
  done
done
