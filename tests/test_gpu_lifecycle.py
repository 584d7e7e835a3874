"""Contexts come and go: forty create / upload / render / destroy cycles -- plain contexts, two-member contexts on the one GPU, a
device-projected frame, a streamed hand-over, a program set -- must leave the device's free memory and the process's resident set
where they were (rxr_destroy frees every pool, the page-locked staging memory, the events and the streams of a context)."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent('''
    import ctypes as C, os, sys
    sys.path.insert(0, %r)
    import numpy as np
    import torch
    import rusterix_amd
    from rusterix_amd import scenes

    prod = rusterix_amd.load()
    host = prod.lib
    host.rxh_set_device_projection.argtypes = [C.c_int]

    def rss_mb():
        for line in open("/proc/self/status"):
            if line.startswith("VmRSS:"):
                return int(line.split()[1]) / 1024.0
        return 0.0

    def cycle(k):
        # every cycle ends with another KIND of context than it started with: the host mirror destroys the old one and creates the new
        if k %% 2:
            host.rxh_set_devices((C.c_int * 2)(0, 0), 2)
        else:
            host.rxh_set_device(0)
        host.rxh_set_device_projection(1 if k %% 4 >= 2 else 0)
        os.environ["RXR_STREAM_UPLOAD"] = "force" if k %% 3 == 0 else "0"
        got = scenes.render(scenes.map_scene(prod, width=640, height=360, logo_size=64, n_lights=4))
        assert int(got[..., 3].min()) == 255
        got = scenes.render(scenes.box_grid_scene(prod, n=24, width=640, height=360, shader=(k %% 5 == 0)))
        assert int(got[..., :3].max()) > 0

    for k in range(6):   # warm-up: the runtime's own pools, code objects, the first compile
        cycle(k)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    rss0 = rss_mb()
    for k in range(40):
        cycle(k)
    host.rxh_set_device(0)          # (drops the last two-member context)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    rss1 = rss_mb()
    print("LIFECYCLE device_mb", round((free0 - free1) / 2**20, 1), "rss_mb", round(rss1 - rss0, 1))
''') % ROOT


def test_create_render_destroy_cycles_leak_nothing():
    env = dict(os.environ, RXR_SHADER_JIT="0")
    pr = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert pr.returncode == 0, pr.stdout[-2000:] + pr.stderr[-4000:]
    line = [l for l in pr.stdout.splitlines() if l.startswith("LIFECYCLE")][-1].split()
    device_mb, rss_mb = float(line[2]), float(line[4])
    # (a context of these frames holds ~60 MB on the device and ~40 MB of page-locked memory: forty leaked ones would be gigabytes)
    assert device_mb < 96.0, f"device memory grew by {device_mb} MB over 40 context cycles"
    assert rss_mb < 256.0, f"resident set grew by {rss_mb} MB over 40 context cycles"
