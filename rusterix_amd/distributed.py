"""Multi-GPU host for the rasterizer path: one process per GPU, framebuffer sharded by tile rows,
assembled with one RCCL all-gather over xGMI (torch.distributed backend "nccl" == RCCL on ROCm).

The reference has no distributed code; what it does have is the observation this module builds on:
tiles are rendered independently and only concatenated at the end
(reference src/rasterizer.rs:273-275, 559-579).

Sharding: the frame is cut into stripes of RXR_TILE_H (16) rows; stripe s belongs to rank s % world
(interleaved, so that the expensive bottom-of-frame stripes and the cheap sky stripes are spread over
all ranks).  Every rank renders its stripes into a compact [stripes_per_rank*16, W] buffer, the
buffers are all-gathered, and one strided copy puts the stripes back in frame order.
"""
from __future__ import annotations

import numpy as np

TILE_H = 16


def stripes_per_rank(height: int, world: int) -> int:
    n_stripes = (height + TILE_H - 1) // TILE_H
    return (n_stripes + world - 1) // world


def stripe_rows(height: int, world: int, rank: int):
    """Frame rows (start, stop) of every stripe owned by `rank`, in local order."""
    out = []
    n_stripes = (height + TILE_H - 1) // TILE_H
    for s in range(rank, n_stripes, world):
        out.append((s * TILE_H, min((s + 1) * TILE_H, height)))
    return out


def assemble_numpy(gathered: np.ndarray, height: int, width: int, world: int) -> np.ndarray:
    """gathered: [world, stripes_per_rank*16, width, 4] -> frame [height, width, 4] (CPU reference
    of the device-side de-interleave; used by the gloo tests)."""
    spr = stripes_per_rank(height, world)
    g = gathered.reshape(world, spr, TILE_H, width, 4)
    frame = np.ascontiguousarray(g.transpose(1, 0, 2, 3, 4)).reshape(spr * world * TILE_H, width, 4)
    return frame[:height]


def assemble_torch(gathered, height: int, width: int, world: int, out=None):
    """Device-side de-interleave of the all-gathered stripes (one strided copy)."""
    spr = stripes_per_rank(height, world)
    g = gathered.view(world, spr, TILE_H, width, 4).permute(1, 0, 2, 3, 4)
    if out is None:
        return g.reshape(spr * world * TILE_H, width, 4)[:height]
    out.view(spr, world, TILE_H, width, 4).copy_(g)
    return out.view(spr * world * TILE_H, width, 4)[:height]


def extract_stripes(frame: np.ndarray, world: int, rank: int) -> np.ndarray:
    """CPU model of rxr_render_stripes_to: the compact [stripes_per_rank*16, W, 4] buffer rank `rank` owns."""
    height, width = frame.shape[0], frame.shape[1]
    spr = stripes_per_rank(height, world)
    out = np.zeros((spr * TILE_H, width, 4), np.uint8)
    for j, (a, b) in enumerate(stripe_rows(height, world, rank)):
        out[j * TILE_H:j * TILE_H + (b - a)] = frame[a:b]
    return out


class StripeGather:
    """Owns the per-rank stripe buffer(s) and the gathered / assembled frame buffers (double-buffered)
    and performs the exchange step: all_gather_into_tensor (RCCL on GPUs, gloo in the CPU tests) followed
    by the de-interleave copy."""

    def __init__(self, height: int, width: int, world: int, rank: int, device, nbuf: int = 2):
        import torch

        self.h, self.w, self.world, self.rank = height, width, world, rank
        self.spr = stripes_per_rank(height, world)
        rows = self.spr * TILE_H
        self.bands = [torch.zeros((rows, width, 4), dtype=torch.uint8, device=device) for _ in range(nbuf)]
        self.gathered = [torch.zeros((world * rows, width, 4), dtype=torch.uint8, device=device) for _ in range(nbuf)]
        self.frames = [torch.zeros((self.spr * world * TILE_H, width, 4), dtype=torch.uint8, device=device) for _ in range(nbuf)]
        self.nbuf = nbuf

    def band(self, i):
        return self.bands[i % self.nbuf]

    def exchange(self, i):
        """Gathers band(i) from every rank and returns the assembled frame (height x width x 4)."""
        self.exchange_begin(i)
        return self.exchange_end(i)

    # split form for software pipelining: begin(i) queues the collective behind the work already on the
    # current stream (the render of frame i) and returns at once; end(i) makes the current stream wait
    # for it and de-interleaves.  Rendering frame i+1 between the two overlaps it with the gather of
    # frame i (the collective runs on the backend's own stream).
    def exchange_begin(self, i):
        import torch.distributed as dist

        b = i % self.nbuf
        if self.world > 1:
            self._work = getattr(self, "_work", {})
            self._work[b] = dist.all_gather_into_tensor(self.gathered[b], self.bands[b], async_op=True)
        else:
            self.gathered[b].copy_(self.bands[b])

    def exchange_end(self, i):
        b = i % self.nbuf
        if self.world > 1:
            self._work.pop(b).wait()
        return assemble_torch(self.gathered[b], self.h, self.w, self.world, out=self.frames[b])
