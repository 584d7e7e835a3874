"""The RCCL calls of the N > 1 exchange, executed on the one GPU there is: a ONE-rank "nccl" process group (two ranks on one
device are refused by RCCL) and StripeGather(self_collective=True), which then issues the same dist.gather / all_gather_into_tensor
calls -- same tensor shapes, gather lists, communicators, async work handles and stream ordering -- that a multi-GPU run issues,
instead of the world == 1 shortcut copy.  It cannot show link behaviour; it does show that the backend accepts every call the path
makes (the gloo tests exercise the host-staged branch only) and that the stripes rendered through the C ABI come back assembled
byte for byte.  Runs in a child process: the process group must not outlive the test."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = textwrap.dedent('''
    import ctypes as C, datetime, os, sys
    sys.path.insert(0, %r)
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    import socket
    with socket.socket() as _s:  # a free port (the rendezvous of a one-rank group still binds one)
        _s.bind(("127.0.0.1", 0))
        os.environ["MASTER_PORT"] = str(_s.getsockname()[1])
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0), timeout=datetime.timedelta(seconds=120))
    import rusterix_amd
    from rusterix_amd import distributed as D, scenes
    prod = rusterix_amd.load()
    host, rxr = prod.lib, rusterix_amd.rxr_abi()
    host.rxh_set_device(0)
    W, H = 640, 360
    cfg = scenes.map_scene(prod, width=W, height=H, n_lights=4)
    rast = cfg.setup()
    assert host.rxh_rasterizer_upload(rast._h, cfg.scene._h, W, H, cfg.tile_size, cfg.assets._h) == 0
    ctx = C.c_void_p(host.rxh_context())
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    sptr = C.c_void_p(stream.cuda_stream)
    full = torch.empty((H, W, 4), dtype=torch.uint8, device="cuda")
    assert rxr.rxr_render_rows_to(ctx, 0, H, C.c_void_p(full.data_ptr()), sptr) == 0
    assert rxr.rxr_synchronize(ctx) == 0
    ref = full.cpu()
    assert int(ref.max()) > 0
    done = []
    for mode, comms, bucket, depth in (("gather", 1, 1, 1), ("allgather", 1, 1, 1), ("rotate", 3, 1, 2), ("gather", 1, 3, 1), ("rotate", 2, 4, 2), ("allgather", 2, 2, 1)):
        g = D.StripeGather(H, W, 1, 0, device="cuda", nbuf=depth + 1, mode=mode, comms=comms, bucket=bucket, self_collective=True)
        assert g.self_collective and len(g.groups) == comms
        n_ex = 5
        pending = []
        for i in range(n_ex + depth):
            if i < n_ex:
                for k in range(bucket):  # the frames of exchange i, rendered on the stream the collective is queued behind
                    g._bands[i %% g.nbuf][k].zero_()
                    assert rxr.rxr_render_stripes_to(ctx, 0, 1, C.c_void_p(g.band_ptr(i, k)), sptr) == 0
                g.exchange_begin(i)
                pending.append(i)
            if i >= depth:
                j = pending.pop(0)
                out = g.exchange_end(j)
                assert out is not None
                out = out if bucket > 1 else out[None]
                for k in range(bucket):
                    assert torch.equal(out[k][:H].cpu(), ref), (mode, comms, bucket, j, k)
        assert rxr.rxr_synchronize(ctx) == 0
        done.append((mode, comms, bucket, depth))
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("RCCL_SELF_OK", len(done), dist.is_nccl_available())
''') % ROOT


def test_one_rank_rccl_group_runs_every_collective_of_the_exchange():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    pr = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert pr.returncode == 0, pr.stdout[-2000:] + pr.stderr[-4000:]
    assert "RCCL_SELF_OK 6 True" in pr.stdout, pr.stdout[-2000:]
