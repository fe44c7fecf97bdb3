"""Multi-GPU frames: interleaved screen bands + one exchange per step.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo" in CPU tests).
The scene is replicated; rank r renders the bands b with b % world == r (bands of `band_rows`
scanlines = rows of 8x8 pixel tiles, interleaved so the model's silhouette is spread over all GPUs) into a compact
[rows_r, W] XRGB buffer; a single gather moves the buffers to rank 0, which de-interleaves them
into the frame.  There is no exchange inside a frame.  Nothing here renders: the render callback
is the C ABI's mi355_render_device (or, in CPU tests, the oracle).
"""
from __future__ import annotations

import numpy as np

BAND_ROWS = 8       # a band = one row of the kernels' 8x8 pixel tiles: no tile straddles two GPUs (15-row bands, which split
                    # 1080 evenly, cost 12-22 % per rank in half-empty tiles; 135 bands over 2/4/8 GPUs differ by one band at most)


def rows_of_rank(height: int, band_rows: int, world: int, rank: int) -> int:
    return sum(1 for y in range(height) if (y // band_rows) % world == rank)


def row_map(height: int, band_rows: int, world: int):
    """For every screen row y: (owning rank, row inside that rank's compact buffer)."""
    owner = np.empty(height, np.int64)
    local = np.empty(height, np.int64)
    fill = [0] * world
    for y in range(height):
        r = (y // band_rows) % world
        owner[y], local[y] = r, fill[r]
        fill[r] += 1
    return owner, local


def assemble_numpy(parts, height: int, band_rows: int) -> np.ndarray:
    world = len(parts)
    owner, local = row_map(height, band_rows, world)
    out = np.empty((height, parts[0].shape[1]), parts[0].dtype)
    for y in range(height):
        out[y] = parts[owner[y]][local[y]]
    return out


class _StagedWork:
    """A collective that ran on host mirrors of device buffers: wait() = the collective's wait, then the copy back to the device
    (the `staged` transport of the gatherers below: a backend without device collectives -- gloo -- moving device buffers)."""

    def __init__(self, works, after=None):
        self.works = works if isinstance(works, (list, tuple)) else [works]
        self.after = after

    def wait(self):
        for w in self.works:
            if w is not None:
                w.wait()
        if self.after is not None:
            self.after()
            self.after = None
        return True


def _host_like(torch, t):
    """A host mirror of device tensor `t` (pinned when the device is a GPU)."""
    pin = t.device.type == "cuda"
    return torch.zeros(t.shape, dtype=t.dtype, device="cpu", pin_memory=pin)


class FrameGatherer:
    """Gathers per-rank compact band buffers on rank 0 and de-interleaves them (torch tensors).

    staged=True runs the collective on host mirrors of the buffers (device -> host, gather, host -> device on rank 0): the
    transport of `bench.py --dry-run`, where N ranks share one GPU and the backend is gloo.  Everything else -- buffers, slots,
    pending work, de-interleave -- is the path RCCL takes."""

    def __init__(self, width: int, height: int, device, band_rows: int = BAND_ROWS, group=None, frames: int = 1,
                 collective_when_alone: bool = False, staged: bool = False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.W, self.H, self.band_rows, self.device = width, height, band_rows, device
        self.alone = self.world == 1 and not collective_when_alone   # a lone rank skips the collective (unless asked: hardware checks)
        self.frames = frames            # frames per gather (a batched launch renders several at once)
        self.my_rows = rows_of_rank(height, band_rows, self.world, self.rank)
        self.max_rows = max(rows_of_rank(height, band_rows, self.world, r) for r in range(self.world))
        owner, local = row_map(height, band_rows, self.world)
        # frame row y lives at flat row owner*max_rows + local of the gathered [world, max_rows, W] block
        self.src_rows = torch.as_tensor(owner * self.max_rows + local, device=device)
        # double-buffered so frame k+1 can render while frame k is on the wire
        shape = (self.max_rows, width) if frames == 1 else (frames, self.max_rows, width)
        self.send = [torch.zeros(shape, dtype=torch.int32, device=device) for _ in range(2)]
        self.recv = [torch.zeros((self.world,) + shape, dtype=torch.int32, device=device)
                     if self.rank == 0 else None for _ in range(2)]
        self.pending = [None, None]
        self.staged = staged and not self.alone
        if self.staged:
            self.h_send = [_host_like(torch, t) for t in self.send]
            self.h_recv = [_host_like(torch, t) if t is not None else None for t in self.recv]

    def send_buffer(self, slot: int):
        """Wait until buffer `slot` is free again and return it (rank-local compact band buffer)."""
        if self.pending[slot] is not None:
            self.pending[slot].wait()
            self.pending[slot] = None
        return self.send[slot]

    def gather(self, slot: int, async_op: bool = True):
        """Launch the gather of buffer `slot` to rank 0 (one collective per frame)."""
        if self.alone:
            return None
        if self.staged:
            self.h_send[slot].copy_(self.send[slot])          # (device -> host in stream order: the frames are complete)
            lst = list(self.h_recv[slot].unbind(0)) if self.rank == 0 else None
            w = self.dist.gather(self.h_send[slot], gather_list=lst, dst=0, group=self.group, async_op=async_op)
            back = (lambda: self.recv[slot].copy_(self.h_recv[slot])) if self.rank == 0 else None
            w = _StagedWork(w if async_op else None, back)
            if not async_op:
                w.wait()
            self.pending[slot] = w if async_op else None
            return w
        if self.rank == 0:
            lst = list(self.recv[slot].unbind(0))
            w = self.dist.gather(self.send[slot], gather_list=lst, dst=0, group=self.group, async_op=async_op)
        else:
            w = self.dist.gather(self.send[slot], gather_list=None, dst=0, group=self.group, async_op=async_op)
        self.pending[slot] = w if async_op else None
        return w

    def ingest_bytes_per_step(self) -> int:
        """What rank 0 takes in per step: the other ranks' band buffers."""
        return (self.world - 1) * self.max_rows * self.W * 4 * self.frames

    def frame(self, slot: int):
        """Rank 0: the assembled [H, W] frame of buffer `slot` (waits for its gather)."""
        if self.pending[slot] is not None:
            self.pending[slot].wait()
            self.pending[slot] = None
        if self.rank != 0:
            return None
        if self.frames == 1:
            if self.alone:
                return self.send[slot][: self.H]
            flat = self.recv[slot].view(self.world * self.max_rows, self.W)
            return flat.index_select(0, self.src_rows)
        # [frames, H, W]: frame f's row y lives at recv[owner, f, local]
        if self.alone:
            return self.send[slot][:, : self.H]
        flat = self.recv[slot].permute(1, 0, 2, 3).reshape(self.frames, self.world * self.max_rows, self.W)
        return flat.index_select(1, self.src_rows)

    def drain(self):
        for s in (0, 1):
            if self.pending[s] is not None:
                self.pending[s].wait()
                self.pending[s] = None


class SpreadAssembler:
    """Bands of a step of several frames, assembled WHERE THE FRAMES STAY: frame j of a step belongs to rank j % world, every rank
    sends each frame's bands to the frame's owner and takes in the bands of the frames it owns -- one grouped all-to-all exchange
    per step instead of a gather.  Rank 0 is no funnel: at N ranks a rank takes in (N - 1) / N of frames / N frames over N - 1
    links at once (8 ranks, 8 frames of 3840 x 2160: 29 MB per rank and step against 234 MB into rank 0), and every link carries
    traffic in both directions.  The frames of a step end up spread over the ranks in orbit order (rank r: frames r, r + N, ...):
    what a presenter that takes frames in turn -- or the next stage of a pipeline -- wants anyway; a caller that needs them all on
    one device uses FrameGatherer.

    Same interface as FrameGatherer: send_buffer(slot) -> [frames, max_rows, W] (frame j of the step goes to index
    slot_of_frame(j): the buffer is ordered by destination, so that every peer's chunk is contiguous), gather(slot), frame(slot)
    -> this rank's frames [frames / world, H, W], drain()."""

    def __init__(self, width: int, height: int, device, frames: int, band_rows: int = BAND_ROWS, group=None, staged: bool = False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if frames % self.world:
            raise ValueError("SpreadAssembler: %d frames per step do not divide over %d ranks" % (frames, self.world))
        self.W, self.H, self.band_rows, self.device = width, height, band_rows, device
        self.alone = self.world == 1
        self.frames, self.n_own = frames, frames // self.world
        self.my_rows = rows_of_rank(height, band_rows, self.world, self.rank)
        self.max_rows = max(rows_of_rank(height, band_rows, self.world, r) for r in range(self.world))
        owner, local = row_map(height, band_rows, self.world)
        self.src_rows = torch.as_tensor(owner * self.max_rows + local, device=device)
        self.send = [torch.zeros((frames, self.max_rows, width), dtype=torch.int32, device=device) for _ in range(2)]
        # recv[slot][s] = rank s's bands of the frames this rank owns
        self.recv = [torch.zeros((self.world, self.n_own, self.max_rows, width), dtype=torch.int32, device=device) for _ in range(2)]
        self.pending = [None, None]
        self.staged = staged and not self.alone       # host mirrors for a backend without device collectives (FrameGatherer)
        if self.staged:
            self.h_send = [_host_like(torch, t) for t in self.send]
            self.h_recv = [_host_like(torch, t) for t in self.recv]

    def slot_of_frame(self, j: int) -> int:
        """Index in the send buffer of frame j of the step (frames of one owner side by side)."""
        return (j % self.world) * self.n_own + j // self.world

    def owned(self, step_frames):
        """The frames of a step (in orbit order) this rank ends up with."""
        return list(step_frames)[self.rank::self.world]

    def send_buffer(self, slot: int):
        self._wait(slot)
        return self.send[slot]

    def _wait(self, slot: int):
        if self.pending[slot] is not None:
            for w in self.pending[slot]:
                w.wait()
            self.pending[slot] = None

    def gather(self, slot: int, async_op: bool = True):
        """The step's exchange: every pair of ranks swaps the bands of each other's frames (grouped point-to-point operations:
        one ncclGroup on RCCL)."""
        n = self.n_own
        self.recv[slot][self.rank].copy_(self.send[slot][self.rank * n:(self.rank + 1) * n])
        if self.alone:
            return None
        src, dst = self.send[slot], self.recv[slot]
        if self.staged:
            self.h_send[slot].copy_(self.send[slot])
            src, dst = self.h_send[slot], self.h_recv[slot]
        ops = []
        for s in range(self.world):
            if s == self.rank:
                continue
            peer = s if self.group is None else self.dist.get_global_rank(self.group, s)
            ops.append(self.dist.P2POp(self.dist.isend, src[s * n:(s + 1) * n], peer, self.group))
            ops.append(self.dist.P2POp(self.dist.irecv, dst[s], peer, self.group))
        works = self.dist.batch_isend_irecv(ops)
        if self.staged:
            def back(slot=slot):
                for s in range(self.world):
                    if s != self.rank:
                        self.recv[slot][s].copy_(self.h_recv[slot][s])
            works = [_StagedWork(works, back)]
        if async_op:
            self.pending[slot] = works
        else:
            for w in works:
                w.wait()
        return works

    def frame(self, slot: int):
        """This rank's frames of the step, [frames / world, H, W] (waits for the exchange)."""
        self._wait(slot)
        flat = self.recv[slot].permute(1, 0, 2, 3).reshape(self.n_own, self.world * self.max_rows, self.W)
        return flat.index_select(1, self.src_rows)

    def ingest_bytes_per_step(self) -> int:
        return (self.world - 1) * self.n_own * self.max_rows * self.W * 4

    def drain(self):
        for s in (0, 1):
            self._wait(s)


def frames_of_rank(frames, world: int, rank: int):
    """Whole-frame sharding of a batch: rank r takes every world-th frame (a uniform sample of the orbit segment, so
    the ranks' loads stay alike)."""
    return list(frames)[rank::world]


class BatchGatherer:
    """Throughput mode: a step is a batch of frames, every rank renders WHOLE frames of it (frames_of_rank) with one
    batched launch, and one gather per step moves them to rank 0 in orbit order.  Frames are independent, so the
    per-GPU work is exactly that of a single-GPU step; bands (FrameGatherer) are for the latency of one frame."""

    def __init__(self, width: int, height: int, device, frames_per_rank: int, group=None, collective_when_alone: bool = False):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.alone = self.world == 1 and not collective_when_alone
        self.W, self.H, self.n_local = width, height, frames_per_rank
        self.my_rows = self.max_rows = height
        shape = (frames_per_rank, height, width)
        self.send = [torch.zeros(shape, dtype=torch.int32, device=device) for _ in range(2)]
        self.recv = [torch.zeros((self.world,) + shape, dtype=torch.int32, device=device)
                     if self.rank == 0 and not self.alone else None for _ in range(2)]
        self.pending = [None, None]

    def send_buffer(self, slot: int):
        if self.pending[slot] is not None:
            self.pending[slot].wait()
            self.pending[slot] = None
        return self.send[slot]

    def gather(self, slot: int, async_op: bool = True):
        if self.alone:
            return None
        lst = list(self.recv[slot].unbind(0)) if self.rank == 0 else None
        w = self.dist.gather(self.send[slot], gather_list=lst, dst=0, group=self.group, async_op=async_op)
        self.pending[slot] = w if async_op else None
        return w

    def frame(self, slot: int):
        """Rank 0: the step's frames [n_local * world, H, W] in orbit order (frame j*world + r came from rank r)."""
        if self.pending[slot] is not None:
            self.pending[slot].wait()
            self.pending[slot] = None
        if self.rank != 0:
            return None
        if self.alone:
            return self.send[slot]
        return self.recv[slot].permute(1, 0, 2, 3).reshape(self.n_local * self.world, self.H, self.W)

    def drain(self):
        for s in (0, 1):
            if self.pending[s] is not None:
                self.pending[s].wait()
                self.pending[s] = None
