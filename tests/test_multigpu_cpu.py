"""CPU (gloo) coverage of the N>1 path: band ownership maths, gather, de-interleave."""
import os
import subprocess
import sys

import numpy as np
import pytest

from renderer_amd import multigpu


@pytest.mark.parametrize("H,world", [(1080, 1), (1080, 2), (1080, 4), (1080, 8), (2160, 8), (100, 3), (7, 2)])
def test_row_map_is_a_partition(H, world):
    owner, local = multigpu.row_map(H, multigpu.BAND_ROWS, world)
    counts = [multigpu.rows_of_rank(H, multigpu.BAND_ROWS, world, r) for r in range(world)]
    assert sum(counts) == H
    for r in range(world):
        assert sorted(local[owner == r].tolist()) == list(range(counts[r]))
    if H % (multigpu.BAND_ROWS * world) == 0:
        assert len(set(counts)) == 1          # 1080 and 2160 split evenly over 1/2/4/8 GPUs


def test_assemble_numpy_roundtrip():
    H, W, world = 97, 13, 3
    full = np.arange(H * W, dtype=np.uint32).reshape(H, W)
    owner, _ = multigpu.row_map(H, multigpu.BAND_ROWS, world)
    parts = [full[owner == r] for r in range(world)]
    assert np.array_equal(multigpu.assemble_numpy(parts, H, multigpu.BAND_ROWS), full)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_gloo_band_gather(tmp_path, world):
    """(world 8: the layout of BASELINE config 5 -- eight ranks, a frame of 2 x 8 x 8 + 5 scanlines so that the ranks own
    different numbers of rows and the last band is short)"""
    out = tmp_path / "result.txt"
    port = 29500 + (os.getpid() % 2000) + world
    size = ["320", "180", "3"] if world < 8 else ["256", "133", "2"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(os.path.dirname(__file__), "_gloo_band_worker.py"), str(out)] + size
    env = dict(os.environ, OMP_NUM_THREADS="2")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert out.read_text() == "OK"


def test_gloo_band_gather_of_batched_frames(tmp_path):
    """A batched launch hands the gatherer several frames per step: [frames, rows, W] buffers, one gather, per-frame
    de-interleave."""
    out = tmp_path / "result.txt"
    port = 29500 + (os.getpid() % 2000) + 7
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(os.path.dirname(__file__), "_gloo_band_worker.py"), str(out), "320", "180", "2", "3"]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert out.read_text() == "OK"


def test_gloo_whole_frame_sharding_of_a_batch(tmp_path):
    """Throughput mode: every rank renders whole frames of the batch (every world-th one), one gather per step puts them
    back in orbit order on rank 0."""
    out = tmp_path / "result.txt"
    port = 29500 + (os.getpid() % 2000) + 11
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(os.path.dirname(__file__), "_gloo_band_worker.py"), str(out), "160", "90", "2", "4", "frames"]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert out.read_text() == "OK"


@pytest.mark.parametrize("world,batch,size", [(2, 4, ("160", "90")), (4, 4, ("128", "77")), (8, 8, ("128", "133"))])
def test_gloo_frames_assembled_on_their_owners(tmp_path, world, batch, size):
    """SpreadAssembler: a step of `batch` frames cut into bands over `world` ranks, frame j assembled on rank j % world by one
    all-to-all exchange (no funnel into rank 0); every rank checks the frames it ends up with against the unsharded render.
    (world 8 with 8 frames = the layout bench.py uses for BASELINE config 5; ragged last band.)"""
    out = tmp_path / "result.txt"
    port = 29500 + (os.getpid() % 2000) + 20 + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(os.path.dirname(__file__), "_gloo_band_worker.py"), str(out), size[0], size[1], "2", str(batch), "spread"]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert out.read_text() == "OK"


@pytest.mark.parametrize("kind", ["rank0", "spread"])
def test_gloo_host_staged_transport(tmp_path, kind):
    """`bench.py --dry-run` moves device buffers through host mirrors (multigpu `staged=True`: device -> host, the collective on
    the mirrors, host -> device in the work's wait): the same gatherers, the same slots and pending work, frames identical."""
    out = tmp_path / "result.txt"
    port = 29500 + (os.getpid() % 2000) + 40 + (1 if kind == "spread" else 0)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(os.path.dirname(__file__), "_gloo_band_worker.py"), str(out), "160", "90", "2", "4"] + (["spread"] if kind == "spread" else [])
    env = dict(os.environ, OMP_NUM_THREADS="2", MI355_TEST_STAGED="1")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert out.read_text() == "OK"
