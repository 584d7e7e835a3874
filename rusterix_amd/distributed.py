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


def assemble_bucket_torch(gathered, height: int, width: int, world: int, out):
    """Device-side de-interleave of a gathered bucket: gathered [world, K, spr*16, width, 4] (rank r's K stripe buffers at [r])
    -> out [K, spr*world*16, width, 4] in frame order (one strided copy); returns out[:, :height]."""
    spr = stripes_per_rank(height, world)
    K = gathered.shape[1]
    g = gathered.view(world, K, spr, TILE_H, width, 4).permute(1, 2, 0, 3, 4, 5)  # [K, spr, world, 16, W, 4]
    out.view(K, spr, world, TILE_H, width, 4).copy_(g)
    return out[:, :height]


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
    (`nbuf` of each) and performs the exchange step.

    mode "gather"    : dist.gather to `root` (default; only the root assembles and owns the frame)
    mode "allgather" : dist.all_gather_into_tensor (every rank assembles the frame)
    mode "rotate"    : dist.gather to rank (i mod world) for exchange i: every frame is still gathered whole, but the root -- whose
                       N-1 incoming links bound the fixed-root gather -- changes from exchange to exchange, so all N*(N-1) directed
                       xGMI links carry traffic (a consumer per GPU: encoders, displays, N-way multi-view)

    bucket = K > 1   : ONE exchange moves the stripes of K consecutive frames (fewer, larger collectives: a rank's share of a 4K
                       frame is 25 us of GPU work, less than the host spends on issuing one collective).  The unit of band(i),
                       exchange_begin(i), exchange_end(i) and root_of(i) is then the bucket: band(i) is [K, rows, W, 4] (frame k of
                       the bucket at band(i)[k]) and exchange_end returns [K, height, W, 4].  bucket = 1 keeps the per-frame shapes.
    """

    def __init__(self, height: int, width: int, world: int, rank: int, device, nbuf: int = 2, mode: str = "gather", root: int = 0,
                 host_staged: bool = False, comms: int = 1, bucket: int = 1, groups=None, self_collective: bool = False):
        import torch
        import torch.distributed as dist

        # comms > 1: that many process groups (communicators) over all ranks, exchange i's collective on group i mod comms.  One
        # communicator runs its collectives one after the other on its own stream; with rotating roots that leaves all links but
        # the current root's idle.  Several communicators let consecutive exchanges' gathers -- to DIFFERENT roots, over disjoint
        # links -- be in flight together (the caller keeps `depth` exchanges between exchange_begin and exchange_end; nbuf > depth).
        # Every rank creates the groups in the same order and issues exchange i's collective on the same group: the order the
        # backend needs.  `groups` hands in communicators made earlier (bench.py times several exchange variants in one run and
        # creates every communicator once).  Untested on xGMI hardware (no multi-GPU box in this build environment).
        # self_collective: with world == 1 the exchange is a local copy; this flag sends it through the backend's collectives all the
        # same (a one-rank communicator) -- the only way a one-GPU box can execute the RCCL calls of the N > 1 path at all
        # (tests/test_gpu_rccl_self.py: two ranks on one device are refused by RCCL)
        assert comms >= 1 and bucket >= 1
        self.self_collective = bool(self_collective) and world == 1
        if groups is not None:
            self.groups = list(groups)
        else:
            self.groups = [None] if comms == 1 or (world == 1 and not self.self_collective) else [dist.new_group(list(range(world))) for _ in range(comms)]

        assert mode in ("gather", "allgather", "rotate")
        self.h, self.w, self.world, self.rank, self.mode, self.root = height, width, world, rank, mode, root
        # host_staged: the collective runs on host copies of the bands (a backend without device collectives: bench.py's
        # one-GPU rehearsal over gloo); synchronous, for rehearsals only
        self.host_staged = host_staged
        self.spr = stripes_per_rank(height, world)
        self.bucket = bucket
        rows = self.spr * TILE_H
        self.rows = rows
        self.nbuf = nbuf
        K = bucket
        self._bands = [torch.zeros((K, rows, width, 4), dtype=torch.uint8, device=device) for _ in range(nbuf)]
        self.owns_frame = mode in ("allgather", "rotate") or rank == root or world == 1
        if self.owns_frame:
            # rank r's K x rows stripe rows land at [r] of the gather target
            self.gathered = [torch.zeros((world, K, rows, width, 4), dtype=torch.uint8, device=device) for _ in range(nbuf)]
            self._frames = [torch.zeros((K, self.spr * world * TILE_H, width, 4), dtype=torch.uint8, device=device) for _ in range(nbuf)]
            self.slots = [[g[r] for r in range(world)] for g in self.gathered]
        else:
            self.gathered, self._frames, self.slots = None, None, None
        self._work = {}

    @property
    def bands(self):
        return [b[0] for b in self._bands] if self.bucket == 1 else self._bands

    @property
    def frames(self):
        """the assembled frames of every buffer ([rows, W, 4] each at bucket 1, [K, rows, W, 4] otherwise); None where the rank owns none"""
        if self._frames is None:
            return None
        return [f[0] for f in self._frames] if self.bucket == 1 else self._frames

    def band(self, i):
        b = self._bands[i % self.nbuf]
        return b[0] if self.bucket == 1 else b

    def band_ptr(self, i, k=0):
        """device address of frame k's stripe buffer inside exchange i's band (what rxr_render_stripes_to writes)"""
        return self._bands[i % self.nbuf][k].data_ptr()

    def root_of(self, i):
        """the rank on which exchange i is assembled (every rank in all-gather mode: then this names rank 0)"""
        return i % self.world if self.mode == "rotate" else self.root

    def exchange(self, i):
        """Blocking form: returns the assembled frame (height x width x 4; K of them at bucket K) on ranks that own it, else None."""
        self.exchange_begin(i)
        return self.exchange_end(i)

    # split form for software pipelining: begin(i) queues the collective behind the work already on the
    # current stream (the renders of exchange i's frames) and returns at once; end(i) makes the current stream wait
    # for it and de-interleaves.  Rendering the next frames between the two overlaps them with the transfer
    # (the collective runs on the backend's own stream).
    def exchange_begin(self, i):
        import torch.distributed as dist

        b = i % self.nbuf
        if self.world == 1 and not self.self_collective:
            self.gathered[b][0].copy_(self._bands[b])
        elif self.host_staged:
            import torch

            torch.cuda.current_stream().synchronize()  # (the renders were queued on, or ordered before, the current stream)
            mine = self._bands[b].cpu()
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
            self._work[b] = dist.all_gather_into_tensor(self.gathered[b].view(self.world * self.bucket, self.rows, self.w, 4), self._bands[b], group=self.groups[i % len(self.groups)], async_op=True)
        else:
            root = self.root_of(i)
            self._work[b] = dist.gather(self._bands[b], self.slots[b] if self.rank == root else None, dst=root,
                                        group=self.groups[i % len(self.groups)], async_op=True)

    def exchange_end(self, i):
        b = i % self.nbuf
        w = self._work.pop(b, None)
        if w is not None:
            w.wait()
        if not self.owns_frame or (self.mode == "rotate" and self.world > 1 and self.rank != self.root_of(i)):
            return None
        out = assemble_bucket_torch(self.gathered[b], self.h, self.w, self.world, out=self._frames[b])
        return out[0] if self.bucket == 1 else out
