#!/usr/bin/env python3
"""prints the timing / parity columns of tools/run_configs.py output files"""
import json
import sys

for path in sys.argv[1:]:
    for l in open(path):
        if not l.startswith("{"):
            if l.strip():
                print(l.strip()[:200])
            continue
        r = json.loads(l)
        p = r.get("parity")
        print(f"{r['config']:11s} dp={int(r.get('device_projection', False))} setup {r['setup_kernels_us']:7.1f} us  raster {r['raster_kernel_us']:7.1f} us  frame {r['frame_ms_device_resident_no_events']:.4f} ms  "
              f"e2e {r['frame_ms_end_to_end']:.3f} ms" + (f"  parity: {p['differing']} differ, max {p['max_abs_diff']}" if p else ""))
