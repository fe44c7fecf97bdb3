"""Tile culling of raytraced frames (k_tile_select): 8x8-pixel tiles whose camera rays cannot pass through any of the boxes at
the top of the tree are set to black without being traced.  The frame must be the frame: against the oracle, against the same
launch with culling switched off (tune flag 16), for cameras near, inside and looking away from the model, single frames and
batches, float output included."""
import os

import numpy as np
import pytest

import renderer_amd as R

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
NCPU = os.cpu_count() or 1


@pytest.fixture(scope="module")
def scenes():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = R.Scene(R.assets.mesh_path(name))
            cache[name].bvh_create()
        return cache[name]
    return get


def look(eye, at):
    return R.camera(list(map(float, eye)), list(map(float, at)))


@pytest.mark.parametrize("mesh", ["dragon_vis.ply", "chessboard.tri", "statue.ply"])
def test_culled_frames_equal_unculled_frames_for_cameras_anywhere(mesh, scenes):
    s = scenes(mesh)
    rng = np.random.default_rng(17)
    _, lights, n = R.benchmark_frame(0)
    W, H = 480, 270
    cams = [R.benchmark_frame(k)[0] for k in (0, 33, 77, 150)]
    for _ in range(28):
        kind = rng.integers(0, 4)
        if kind == 0:      # far away, model small or off screen
            eye = rng.normal(size=3) * 6; at = rng.normal(size=3) * (2 if rng.random() < 0.5 else 0.2)
        elif kind == 1:    # inside the model's box
            eye = rng.uniform(-0.5, 0.5, 3); at = rng.uniform(-1, 1, 3)
        elif kind == 2:    # close, grazing
            eye = rng.normal(size=3); eye = eye / np.linalg.norm(eye) * rng.uniform(1.0, 2.0); at = eye + rng.normal(size=3)
        else:              # looking away
            eye = rng.normal(size=3) * 3; at = eye * 2
        if np.linalg.norm(np.cross(at - eye, [0, 0, 1])) < 1e-3:
            continue
        cams.append(look(eye, at))
    culled_something = 0
    for i, cam in enumerate(cams):
        for mode in (9, 10) if i % 5 == 0 else (9,):
            a = s.render(mode, cam, lights, n, R.default_opts(W, H), want_f32=True)
            b = s.render(mode, cam, lights, n, R.default_opts(W, H, tune=R.tune(nocull=1)), want_f32=True)
            assert np.array_equal(a[0], b[0]), "camera %d mode %d: %d pixels differ" % (i, mode, (a[0] != b[0]).sum())
            assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), "camera %d mode %d: float frame" % (i, mode)
            culled_something += int((a[0] == 0).mean() > 0.3)
    assert culled_something > 4


@pytest.mark.parametrize("mesh,W,H,frame,kw", [
    ("dragon_vis.ply", 1920, 1080, 0, {}),
    ("dragon_vis.ply", 1000, 700, 120, dict(max_ray_depth=1)),
    ("chessboard.tri", 1280, 720, 60, {}),
    ("statue.ply", 333, 187, 9, {}),
    ("dragon_vis.ply", 64, 48, 3, {}),
    ("dragon_vis.ply", 7, 5, 3, {}),
])
def test_culled_frames_equal_the_oracle(oracle, oracle_scene, scenes, mesh, W, H, frame, kw):
    s, osc = scenes(mesh), oracle_scene(mesh, True)
    cam, lights, n = R.benchmark_frame(frame)
    ocam, olights, on = oracle.benchmark_frame(frame)
    g = s.render(9, cam, lights, n, R.default_opts(W, H, **kw), want_f32=True)
    o = osc.render(9, ocam, olights, on, oracle.default_opts(W, H, threads=NCPU, **kw), want_f32=True)
    assert np.array_equal(g[0], o[0])
    assert np.array_equal(g[1].view(np.uint32), o[1].view(np.uint32))


def test_culled_batches_and_stale_buffers(scenes):
    """frames of a batch have different tile lists; the output buffers hold garbage before the launch"""
    s = scenes("dragon_vis.ply")
    W, H = 640, 360
    ks = [0, 40, 80, 120, 160, 199, 10, 20]
    cams = [R.benchmark_frame(k)[0] for k in ks]
    ls = [R.benchmark_frame(k)[1] for k in ks]
    buf = torch.full((len(ks), H, W), 0x55aa55, dtype=torch.int32, device="cuda")
    s.render_batch_device(9, cams, ls, 1, R.default_opts(W, H), [buf[j].data_ptr() for j in range(len(ks))], W * 4, None, 0)
    torch.cuda.synchronize()
    got = buf.cpu().numpy().view(np.uint32)
    for j, k in enumerate(ks):
        ref = s.render(9, cams[j], ls[j], 1, R.default_opts(W, H, tune=R.tune(nocull=1)))[0]
        assert np.array_equal(got[j], ref), "frame %d of the batch" % k


@pytest.mark.parametrize("compact", [0, 1])
@pytest.mark.parametrize("band_rows,count", [(8, 2), (8, 3), (16, 4), (24, 5), (8, 8)])
def test_culled_bands_equal_unculled_bands(scenes, band_rows, count, compact):
    """band sharding (the multi-GPU layout): the tile mask is over the rank's own rows"""
    s = scenes("dragon_vis.ply")
    W, H = 640, 360
    cam, lights, n = R.benchmark_frame(25)
    whole = s.render(9, cam, lights, n, R.default_opts(W, H, tune=R.tune(nocull=1)))[0]
    for b in range(count):
        kw = dict(band_rows=band_rows, band_index=b, band_count=count, compact_rows=compact)
        a = s.render(9, cam, lights, n, R.default_opts(W, H, **kw), want_f32=True)
        rows = [y for y in range(H) if (y // band_rows) % count == b]
        got = a[0] if compact else a[0][rows]
        assert np.array_equal(got, whole[rows]), "band %d of %d" % (b, count)
        c = s.render(9, cam, lights, n, R.default_opts(W, H, tune=R.tune(nocull=1), **kw), want_f32=True)
        assert np.array_equal(a[0], c[0]) and np.array_equal(a[1].view(np.uint32), c[1].view(np.uint32))
