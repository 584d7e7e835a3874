for l in 0 1 2; do RXR_MIN_KERNEL_LEVEL=$l timeout 300 python tools/run_configs.py --configs C5s,C5 --oracle none --frames 20 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('level $l', d['config'], 'raster_us', d['raster_kernel_us'])
"; done
