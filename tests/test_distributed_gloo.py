"""The N>1 path on CPU: world_size-2 (and 3) gloo process groups exercise the stripe partition, the
exchange (gather to rank 0, and the all-gather variant) and the de-interleave exactly as bench.py
runs them over RCCL -- including the begin/end software pipelining; the per-rank render is modelled by
slicing a frame rendered by the CPU oracle (tests only)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rusterix_amd import distributed as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, frame, result_dir, mode, comms=1, depth=1, bucket=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        h, w = frame.shape[0], frame.shape[1]
        g = D.StripeGather(h, w, world, rank, device="cpu", mode=mode, comms=comms, nbuf=depth + 1, bucket=bucket)
        ok = True
        n = 5 if depth == 1 else 11

        def want(i, k=0):  # frame k of exchange i
            return np.roll(frame, (i * bucket + k) * 5, axis=1)

        def fill(i):
            for k in range(bucket):
                (g.band(i) if bucket == 1 else g.band(i)[k]).copy_(torch.from_numpy(D.extract_stripes(want(i, k), world, rank)))

        def check(i, out):
            nonlocal ok
            if g.owns_frame and (mode != "rotate" or rank == g.root_of(i)):  # (rotate: exchange i lives on rank i mod world only)
                ok &= out is not None
                if out is not None:
                    ok &= tuple(out.shape) == ((h, w, 4) if bucket == 1 else (bucket, h, w, 4))
                    for k in range(bucket):
                        ok &= np.array_equal((out if bucket == 1 else out[k]).numpy(), want(i, k))
            else:
                ok &= out is None

        # the same pipelined loop bench.py runs: render(i); begin(i); end(i - depth)
        for i in range(n):
            fill(i)
            g.exchange_begin(i)
            if i >= depth:
                check(i - depth, g.exchange_end(i - depth))
        for i in range(max(0, n - depth), n):
            check(i, g.exchange_end(i))
        # and the blocking form
        fill(0)
        check(0, g.exchange(0))
        np.save(os.path.join(result_dir, f"ok{rank}.npy"), np.array([ok, g.owns_frame]))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,size,mode", [(2, (360, 640), "gather"), (3, (203, 77), "gather"), (2, (16, 32), "gather"),
                                             (2, (100, 64), "allgather"), (3, (120, 48), "rotate"), (2, (64, 40), "rotate")])
def test_stripe_exchange_roundtrip(tmp_path, oracle, world, size, mode):
    from rusterix_amd import scenes

    h, w = size
    frame = scenes.render(scenes.map_scene(oracle, width=w, height=h, logo_size=16, n_lights=1)).copy()
    mp.spawn(_worker, args=(world, _free_port(), frame, str(tmp_path), mode), nprocs=world, join=True)
    owners = 0
    for r in range(world):
        ok, owns = np.load(tmp_path / f"ok{r}.npy")
        assert ok, f"rank {r} assembled a wrong frame"
        owners += int(owns)
    assert owners == (world if mode in ("allgather", "rotate") else 1)


@pytest.mark.parametrize("world,comms,depth,mode", [(2, 2, 2, "rotate"), (3, 3, 3, "rotate"), (3, 2, 2, "gather")])
def test_several_communicators_and_a_deeper_pipeline(tmp_path, oracle, world, comms, depth, mode):
    """frame i's collective on communicator i mod comms, `depth` frames in flight between exchange_begin and exchange_end
    (depth + 1 buffers): every frame still arrives whole on its root"""
    from rusterix_amd import scenes

    frame = scenes.render(scenes.map_scene(oracle, width=56, height=100, logo_size=16, n_lights=1)).copy()
    mp.spawn(_worker, args=(world, _free_port(), frame, str(tmp_path), mode, comms, depth), nprocs=world, join=True)
    for r in range(world):
        ok, _ = np.load(tmp_path / f"ok{r}.npy")
        assert ok, f"rank {r} assembled a wrong frame"


def test_partition_covers_every_row_once():
    for h in (1, 15, 16, 17, 360, 2160, 4320):
        for world in (1, 2, 3, 4, 8):
            seen = np.zeros(h, np.int32)
            for r in range(world):
                rows = D.stripe_rows(h, world, r)
                assert len(rows) <= D.stripes_per_rank(h, world)
                for a, b in rows:
                    seen[a:b] += 1
            assert (seen == 1).all()


def test_assemble_numpy_matches_torch():
    rng = np.random.default_rng(1)
    h, w, world = 100, 24, 4
    frame = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    parts = np.concatenate([D.extract_stripes(frame, world, r) for r in range(world)], axis=0)
    a = D.assemble_numpy(parts, h, w, world)
    b = D.assemble_torch(torch.from_numpy(parts), h, w, world).numpy()
    out = torch.zeros((D.stripes_per_rank(h, world) * world * D.TILE_H, w, 4), dtype=torch.uint8)
    c = D.assemble_torch(torch.from_numpy(parts), h, w, world, out=out).numpy()
    assert np.array_equal(a, frame) and np.array_equal(b, frame) and np.array_equal(c, frame)


def test_single_process_world_one():
    frame = np.random.default_rng(2).integers(0, 256, (50, 20, 4), dtype=np.uint8)
    g = D.StripeGather(50, 20, 1, 0, device="cpu")
    g.band(0).copy_(torch.from_numpy(D.extract_stripes(frame, 1, 0)))
    assert np.array_equal(g.exchange(0).numpy(), frame)


@pytest.mark.parametrize("world,bucket,comms,depth,mode", [(2, 3, 1, 1, "gather"), (3, 2, 2, 2, "rotate"), (2, 4, 1, 1, "allgather")])
def test_buckets_of_frames_per_exchange(tmp_path, oracle, world, bucket, comms, depth, mode):
    """bucket = K: one collective moves the stripes of K frames (band(i) is [K, rows, W, 4]) and the root de-interleaves K whole frames"""
    from rusterix_amd import scenes

    frame = scenes.render(scenes.map_scene(oracle, width=56, height=90, logo_size=16, n_lights=1)).copy()
    mp.spawn(_worker, args=(world, _free_port(), frame, str(tmp_path), mode, comms, depth, bucket), nprocs=world, join=True)
    for r in range(world):
        ok, _ = np.load(tmp_path / f"ok{r}.npy")
        assert ok, f"rank {r} assembled a wrong frame"
