#!/bin/bash
# A-B-A-B of the in-tree library against build/variants/librxr_hip_prev.so: bench line + configurations (host- and device-projected).
cd "$(dirname "$0")/.."
cp rusterix_amd/csrc/librxr_hip.so /tmp/new.so
trap 'cp /tmp/new.so rusterix_amd/csrc/librxr_hip.so' EXIT   # the product library comes back on ANY exit
for r in 1 2; do
for name in new prev; do
  if [ $name = prev ]; then cp build/variants/librxr_hip_prev.so rusterix_amd/csrc/librxr_hip.so; else cp /tmp/new.so rusterix_amd/csrc/librxr_hip.so; fi
  timeout 120 python bench.py --steps 300 --warmup 30 --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$name bench ms', d['ms_per_step'], 'kernel', d['roofline']['kernel_avg_us'], 'setup', d['roofline']['setup_kernels_avg_us'])"
  for dp in "" "--device-projection"; do
  timeout 300 python tools/run_configs.py --configs ${1:-C5,C3} --oracle none --frames 20 $dp 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$name $dp', d['config'], 'setup_us', d['setup_kernels_us'], 'raster_us', d['raster_kernel_us'], 'frame_ms', d['frame_ms_device_resident_no_events'])
"
  done
done; done
cp /tmp/new.so rusterix_amd/csrc/librxr_hip.so
