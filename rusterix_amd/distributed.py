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
