"""GPU parity of the raytracer's compile-time extras (Raytracer.cc:70-80), off by default like in
the reference: refractions and ray-cast ambient occlusion.  The oracle's restatement of both is pinned to the REAL
reference built with -DREFRACTIONS / -DAMBIENT_OCCLUSION (tests/test_refcore_pins.py); here the HIP path, through the
C ABI, must give the oracle's frames: identical XRGB words and identical floats."""
import os

import numpy as np
import pytest

import renderer_amd as R

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
NCPU = os.cpu_count() or 1


@pytest.fixture(scope="module")
def gpu_scene():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = R.Scene(R.assets.mesh_path(name))
            cache[name].bvh_create()
        return cache[name]
    return get


def frames(oracle, oracle_scene, gpu_scene, mesh, mode, W, H, frame, two, gpu_kw, orc_kw):
    hs, osc = gpu_scene(mesh), oracle_scene(mesh, True)
    cam, lights, n = R.benchmark_frame(frame, two)
    ocam, olights, on = oracle.benchmark_frame(frame, two)
    g = hs.render(mode, cam, lights, n, R.default_opts(W, H, **gpu_kw), want_f32=True)
    o = osc.render(mode, ocam, olights, on, oracle.default_opts(W, H, threads=NCPU, **orc_kw), want_f32=True)
    return g, o


def same(g, o):
    assert int((g[0] != o[0]).sum()) == 0, "%d pixels differ from the oracle" % int((g[0] != o[0]).sum())
    assert np.array_equal(g[1].view(np.uint32), o[1].view(np.uint32)), "float frame differs (max %g)" % float(np.abs(g[1] - o[1]).max())


@pytest.mark.parametrize("mesh,W,H,frame,two,kw", [
    ("chessboard.tri", 800, 600, 0, True, {}),
    ("chessboard.tri", 1920, 1080, 30, False, {}),
    ("dragon_vis.ply", 800, 600, 60, False, {}),
    ("dragon_vis.ply", 1920, 1080, 0, True, {}),
    ("statue.ply", 800, 600, 120, False, dict(max_ray_depth=4)),
    ("dragon_vis.ply", 640, 360, 10, False, dict(max_ray_depth=2)),
    ("dragon_vis.ply", 640, 360, 10, False, dict(max_ray_depth=1)),
    ("chessboard.tri", 640, 360, 90, False, dict(use_reflections=0)),               # refracted rays only
    ("chessboard.tri", 640, 360, 90, True, dict(use_shadows=0)),
    ("dragon_vis.ply", 333, 187, 5, False, {}),                                      # ragged tiles
], ids=["chess_800", "chess_1080p", "dragon_800", "dragon_1080p_2lights", "statue_depth4", "depth2", "depth1", "no_reflections",
        "no_shadows", "ragged"])
def test_refractions_match_the_oracle(oracle, oracle_scene, gpu_scene, mesh, W, H, frame, two, kw):
    kw = dict(kw, use_refractions=1)
    g, o = frames(oracle, oracle_scene, gpu_scene, mesh, 9, W, H, frame, two, kw, kw)
    same(g, o)
    plain = gpu_scene(mesh).render(9, *R.benchmark_frame(frame, two), R.default_opts(W, H, **dict(kw, use_refractions=0)))
    if kw.get("max_ray_depth", 3) > 1:
        assert (plain[0] != g[0]).sum() > W * H // 200          # (the option changes the picture)


def test_refractions_antialiased_and_foreign_walk_order(oracle, oracle_scene, gpu_scene):
    kw = dict(use_refractions=1)
    g, o = frames(oracle, oracle_scene, gpu_scene, "dragon_vis.ply", 10, 400, 300, 20, False, kw, dict(kw, antialias=1))
    same(g, o)
    # the reference's fixed walk order (tune flag 4) and the exact box test (flag 1) give the same frame
    for flags in (4, 1, 5):
        t = [0] * 8; t[5] = flags
        g2, _ = frames(oracle, oracle_scene, gpu_scene, "dragon_vis.ply", 10, 400, 300, 20, False, dict(kw, tune=tuple(t)), dict(kw, antialias=1))
        assert np.array_equal(g2[0], g[0]) and np.array_equal(g2[1].view(np.uint32), g[1].view(np.uint32))


@pytest.mark.parametrize("mesh,W,H,frame,kw", [
    ("chessboard.tri", 640, 360, 0, {}),
    ("dragon_vis.ply", 800, 600, 45, {}),
    ("statue.ply", 480, 270, 100, dict(ao_samples=8, ao_range=0.4)),
    ("dragon_vis.ply", 320, 180, 3, dict(use_refractions=1)),                        # both extras at once
], ids=["chessboard", "dragon", "statue_8_samples", "with_refractions"])
def test_raycast_ambient_occlusion_matches_the_oracle(oracle, oracle_scene, gpu_scene, mesh, W, H, frame, kw):
    """Same counter-based generator on both sides (mi355_render.h / oracle.h orc_ao_random): bit-identical frames.  That
    the generator stands in for the reference's rand() without bias is the CPU test in tests/test_refcore_pins.py."""
    g, o = frames(oracle, oracle_scene, gpu_scene, mesh, 9, W, H, frame, False, dict(kw, ambient_occlusion=1), dict(kw, ambient_occlusion=2))
    same(g, o)


def test_extras_do_not_depend_on_banding_or_batching(oracle_scene, gpu_scene):
    """Ambient-occlusion samples are keyed by SCREEN pixel: a frame rendered in bands (the multi-GPU layout) or as part of
    a batch is the whole frame."""
    hs = gpu_scene("dragon_vis.ply")
    W, H = 640, 360
    cam, lights, n = R.benchmark_frame(12)
    kw = dict(use_refractions=1, ambient_occlusion=1, ao_samples=8)
    whole = hs.render(9, cam, lights, n, R.default_opts(W, H, **kw))[0]
    out = np.zeros_like(whole)
    for b in range(3):
        part = hs.render(9, cam, lights, n, R.default_opts(W, H, band_rows=8, band_index=b, band_count=3, **kw))[0]
        rows = [y for y in range(H) if (y // 8) % 3 == b]
        out[rows] = part[rows]
    assert np.array_equal(out, whole)
    cams = [R.benchmark_frame(k)[0] for k in (12, 13)]
    ls = [R.benchmark_frame(k)[1] for k in (12, 13)]
    buf = torch.zeros((2, H, W), dtype=torch.int32, device="cuda")
    hs.render_batch_device(9, cams, ls, n, R.default_opts(W, H, **kw), [buf[j].data_ptr() for j in range(2)], W * 4, None, 0)
    torch.cuda.synchronize()
    assert np.array_equal(buf[0].cpu().numpy().view(np.uint32), whole)
