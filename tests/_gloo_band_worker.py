"""Worker for test_multigpu_cpu.py: world_size-N gloo run of the band-shard + gather + de-interleave path.

The pixels come from the CPU oracle here (this is a CPU test of the distributed plumbing in
renderer_amd/multigpu.py, which is backend-agnostic); on GPUs the same FrameGatherer moves buffers that
mi355_render_device filled."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle_ctypes as O          # noqa: E402
from renderer_amd import assets, multigpu      # noqa: E402


def main():
    out_path, W, H, frames = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    batch = int(sys.argv[5]) if len(sys.argv) > 5 else 1      # frames per gather (a batched launch on the GPU)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    staged = os.environ.get("MI355_TEST_STAGED") == "1"        # the host-staged transport of `bench.py --dry-run`
    s = O.Scene(assets.mesh_path("dragon_vis.ply"))
    s.bvh_ensure(os.path.join(assets.cache_dir(), "dragon_vis.ply.oracle.bvh"))
    if len(sys.argv) > 6 and sys.argv[6] == "frames":          # whole-frame sharding of a batch (BatchGatherer)
        per_rank = batch // world
        g = multigpu.BatchGatherer(W, H, torch.device("cpu"), per_rank)
        ok = True
        for step in range(frames):
            fs = [step * batch + j for j in range(batch)]
            buf = g.send_buffer(step & 1)
            for j, f in enumerate(multigpu.frames_of_rank(fs, world, rank)):
                cam, lights, n = O.benchmark_frame(f)
                buf[j] = torch.from_numpy(s.render(9, cam, lights, n, O.default_opts(W, H))[0].astype(np.int32))
            g.gather(step & 1)
            if rank == 0:
                got = g.frame(step & 1).numpy().astype(np.uint32)
                ok = ok and got.shape == (batch, H, W)
                for j, f in enumerate(fs):
                    cam, lights, n = O.benchmark_frame(f)
                    ok = ok and bool(np.array_equal(got[j], s.render(9, cam, lights, n, O.default_opts(W, H))[0]))
        g.drain()
        dist.barrier()
        if rank == 0:
            open(out_path, "w").write("OK" if ok else "MISMATCH")
        dist.destroy_process_group()
        return
    if len(sys.argv) > 6 and sys.argv[6] == "spread":          # bands of a step's frames, every frame assembled on its owner (SpreadAssembler)
        g = multigpu.SpreadAssembler(W, H, torch.device("cpu"), frames=batch, staged=staged)
        ys = np.arange(H)
        mine = (ys // multigpu.BAND_ROWS) % world == rank
        ok = g.my_rows == int(mine.sum())
        for step in range(frames):
            fs = [step * batch + j for j in range(batch)]
            buf = g.send_buffer(step & 1)
            buf.zero_()
            for j, f in enumerate(fs):
                cam, lights, n = O.benchmark_frame(f)
                o = O.default_opts(W, H, band_rows=multigpu.BAND_ROWS, band_index=rank, band_count=world)
                buf[g.slot_of_frame(j), : g.my_rows] = torch.from_numpy(s.render(9, cam, lights, n, o)[0][mine].astype(np.int32))
            g.gather(step & 1)                   # async: the next step's frames render meanwhile
            got = g.frame(step & 1).numpy().astype(np.uint32)
            own = g.owned(fs)
            ok = ok and got.shape == (batch // world, H, W) and own == fs[rank::world]
            for m, f in enumerate(own):
                cam, lights, n = O.benchmark_frame(f)
                full = s.render(9, cam, lights, n, O.default_opts(W, H))[0]
                ok = ok and bool(np.array_equal(got[m], full)) and int((full != 0).sum()) > 0
        g.drain()
        flag = torch.tensor([1 if ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)          # every rank checks the frames it owns
        if rank == 0:
            open(out_path, "w").write("OK" if int(flag[0]) == 1 else "MISMATCH")
        dist.destroy_process_group()
        return
    g = multigpu.FrameGatherer(W, H, torch.device("cpu"), frames=batch, staged=staged)
    assert g.my_rows == multigpu.rows_of_rank(H, multigpu.BAND_ROWS, world, rank)
    ok = True
    if batch > 1:
        ys = np.arange(H)
        mine = (ys // multigpu.BAND_ROWS) % world == rank
        for step in range(frames):
            buf = g.send_buffer(step & 1)
            buf.zero_()
            fulls = []
            for j in range(batch):
                cam, lights, n = O.benchmark_frame(step * batch + j)
                o = O.default_opts(W, H, band_rows=multigpu.BAND_ROWS, band_index=rank, band_count=world)
                img, _, _ = s.render(9, cam, lights, n, o)
                buf[j, : g.my_rows] = torch.from_numpy(img[mine].astype(np.int32))
                if rank == 0:
                    fulls.append(s.render(9, cam, lights, n, O.default_opts(W, H))[0])
            g.gather(step & 1)
            if rank == 0:
                got = g.frame(step & 1).numpy().astype(np.uint32)
                ok = ok and got.shape == (batch, H, W) and all(bool(np.array_equal(got[j], fulls[j])) for j in range(batch))
        frames = 0
    for k in range(frames):
        cam, lights, n = O.benchmark_frame(k)
        o = O.default_opts(W, H, band_rows=multigpu.BAND_ROWS, band_index=rank, band_count=world)
        img, _, _ = s.render(9, cam, lights, n, o)
        ys = np.arange(H)
        mine = (ys // multigpu.BAND_ROWS) % world == rank
        buf = g.send_buffer(k & 1)
        buf.zero_()
        buf[: g.my_rows] = torch.from_numpy(img[mine].astype(np.int32))
        g.gather(k & 1)                      # async; next frame renders meanwhile
        if rank == 0:
            full, _, _ = s.render(9, cam, lights, n, O.default_opts(W, H))
            got = g.frame(k & 1).numpy().astype(np.uint32)
            ok = ok and bool(np.array_equal(got, full)) and int((full != 0).sum()) > 0
    g.drain()
    dist.barrier()
    if rank == 0:
        open(out_path, "w").write("OK" if ok else "MISMATCH")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
