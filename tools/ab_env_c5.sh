#!/bin/bash
# A-B-A-B of one environment knob on configuration C5 (tools/run_configs.py, host-projected): usage tools/ab_env_c5.sh NAME value_a value_b [configs]
cd "$(dirname "$0")/.."
N=$1; A=$2; B=$3; CFG=${4:-C5}
for r in 1 2; do for v in $A $B; do
  env $N=$v python tools/run_configs.py --configs $CFG --oracle none --frames 20 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$N=$v', d['config'], 'setup_us', d['setup_kernels_us'], 'raster_us', d['raster_kernel_us'], 'frame_ms', d['frame_ms_device_resident_no_events'], 'e2e_ms', d.get('frame_ms_end_to_end'))
"
done; done
