"""Multi-GPU host for the rasterizer path: one process per GPU, framebuffer sharded by tile rows,
assembled on rank 0 with one RCCL gather over xGMI (torch.distributed backend "nccl" == RCCL on ROCm).

The reference has no distributed code; what it does have is the observation this module builds on:
tiles are rendered independently and only concatenated at the end
(reference src/rasterizer.rs:273-275, 559-579).

Sharding: the frame is cut into stripes of RXR_STRIPE_ROWS (16) rows; stripe s belongs to rank
s % world (interleaved, so that the expensive bottom-of-frame stripes and the cheap sky stripes are
spread over all ranks).  Every rank renders its stripes into a compact [stripes_per_rank*16, W]
buffer (rxr_render_stripes_to); the buffers are gathered to rank 0 -- on a fully connected xGMI node
the N-1 transfers run concurrently, one per link -- and one strided copy on rank 0 puts the stripes
back in frame order.  The exchange is split into begin/end so that the caller can render frame i+1
while frame i is in flight.
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
    """gathered: [world, stripes_per_rank*16, width, 4] -> frame [height, width, 4] (CPU model of the
    device-side de-interleave; used by the tests)."""
    spr = stripes_per_rank(height, world)
    g = gathered.reshape(world, spr, TILE_H, width, 4)
    frame = np.ascontiguousarray(g.transpose(1, 0, 2, 3, 4)).reshape(spr * world * TILE_H, width, 4)
    return frame[:height]


def assemble_torch(gathered, height: int, width: int, world: int, out=None):
    """Device-side de-interleave of the gathered stripes (one strided copy)."""
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
    """Owns the per-rank stripe buffers and, on the root, the gathered / assembled frame buffers
    (all double-buffered) and performs the exchange step.

    mode "gather"    : dist.gather to `root` (default; only the root assembles and owns the frame)
    mode "allgather" : dist.all_gather_into_tensor (every rank assembles the frame)
    mode "rotate"    : dist.gather to rank (i mod world) for frame i: every frame is still gathered whole, but the root -- whose
                       N-1 incoming links bound the fixed-root gather -- changes from frame to frame, so all N*(N-1) directed
                       xGMI links carry traffic (a consumer per GPU: encoders, displays, N-way multi-view)
    """

    def __init__(self, height: int, width: int, world: int, rank: int, device, nbuf: int = 2, mode: str = "gather", root: int = 0,
                 host_staged: bool = False, comms: int = 1):
        import torch
        import torch.distributed as dist

        # comms > 1: that many process groups (communicators) over all ranks, frame i's collective on group i mod comms.  One
        # communicator runs its collectives one after the other on its own stream; with rotating roots that leaves all links but
        # the current root's idle.  Several communicators let consecutive frames' gathers -- to DIFFERENT roots, over disjoint
        # links -- be in flight together (the caller keeps `depth` frames between exchange_begin and exchange_end; nbuf > depth).
        # Every rank creates the groups in the same order and issues frame i's collective on the same group: the order the
        # backend needs.  Untested on xGMI hardware (no multi-GPU box in this build environment); opt-in.
        assert comms >= 1
        self.groups = [None] if comms == 1 or world == 1 else [dist.new_group(list(range(world))) for _ in range(comms)]

        assert mode in ("gather", "allgather", "rotate")
        self.h, self.w, self.world, self.rank, self.mode, self.root = height, width, world, rank, mode, root
        # host_staged: the collective runs on host copies of the bands (a backend without device collectives: bench.py's
        # one-GPU rehearsal over gloo); synchronous, for rehearsals only
        self.host_staged = host_staged
        self.spr = stripes_per_rank(height, world)
        rows = self.spr * TILE_H
        self.nbuf = nbuf
        self.bands = [torch.zeros((rows, width, 4), dtype=torch.uint8, device=device) for _ in range(nbuf)]
        self.owns_frame = mode in ("allgather", "rotate") or rank == root or world == 1
        if self.owns_frame:
            self.gathered = [torch.zeros((world * rows, width, 4), dtype=torch.uint8, device=device) for _ in range(nbuf)]
            self.frames = [torch.zeros((self.spr * world * TILE_H, width, 4), dtype=torch.uint8, device=device) for _ in range(nbuf)]
            # per-source views of the gather target (rank r's stripes land at rows [r*rows, (r+1)*rows))
            self.slots = [[g[r * rows:(r + 1) * rows] for r in range(world)] for g in self.gathered]
        else:
            self.gathered, self.frames, self.slots = None, None, None
        self._work = {}

    def band(self, i):
        return self.bands[i % self.nbuf]

    def root_of(self, i):
        """the rank on which frame i is assembled (every rank in all-gather mode: then this names rank 0)"""
        return i % self.world if self.mode == "rotate" else self.root

    def exchange(self, i):
        """Blocking form: returns the assembled frame (height x width x 4) on ranks that own it, else None."""
        self.exchange_begin(i)
        return self.exchange_end(i)

    # split form for software pipelining: begin(i) queues the collective behind the work already on the
    # current stream (the render of frame i) and returns at once; end(i) makes the current stream wait
    # for it and de-interleaves.  Rendering frame i+1 between the two overlaps it with the transfer of
    # frame i (the collective runs on the backend's own stream).
    def exchange_begin(self, i):
        import torch.distributed as dist

        b = i % self.nbuf
        if self.world == 1:
            self.gathered[b].copy_(self.bands[b])
        elif self.host_staged:
            import torch

            torch.cuda.current_stream().synchronize()  # (the render of frame i was queued on the current stream)
            mine = self.bands[b].cpu()
            if self.mode == "allgather":
                parts = [torch.empty_like(mine) for _ in range(self.world)]
                dist.all_gather(parts, mine, group=self.groups[i % len(self.groups)])
            else:
                root = self.root_of(i)
                parts = [torch.empty_like(mine) for _ in range(self.world)] if self.rank == root else None
                dist.gather(mine, parts, dst=root, group=self.groups[i % len(self.groups)])
            if parts is not None:
                for r in range(self.world):
                    self.slots[b][r].copy_(parts[r])
        elif self.mode == "allgather":
            self._work[b] = dist.all_gather_into_tensor(self.gathered[b], self.bands[b], group=self.groups[i % len(self.groups)], async_op=True)
        else:
            root = self.root_of(i)
            self._work[b] = dist.gather(self.bands[b], self.slots[b] if self.rank == root else None, dst=root,
                                        group=self.groups[i % len(self.groups)], async_op=True)

    def exchange_end(self, i):
        b = i % self.nbuf
        w = self._work.pop(b, None)
        if w is not None:
            w.wait()
        if not self.owns_frame or (self.mode == "rotate" and self.world > 1 and self.rank != self.root_of(i)):
            return None
        return assemble_torch(self.gathered[b], self.h, self.w, self.world, out=self.frames[b])
