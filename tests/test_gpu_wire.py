"""Mode 3, the wireframe (Scene::renderWireframe, Rasterizers.cc:117-187) on the GPU against the oracle: the lines blend over
each other in triangle order, so every pixel depends on that order -- the frames must be identical words.  (The oracle's
line code is a restatement of Wu.cc as read, not pinned to reference output: oracle/oracle.cc says why.)"""
import os

import numpy as np
import pytest

import renderer_amd as R

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.mark.parametrize("mesh,W,H,frame", [
    ("chessboard.tri", 800, 600, 0), ("chessboard.tri", 1920, 1080, 40),
    ("dragon_vis.ply", 800, 600, 100), ("dragon_vis.ply", 1920, 1080, 0),
    ("statue.ply", 640, 360, 170), ("dragon_vis.ply", 333, 187, 5), ("chessboard.tri", 33, 17, 9),
])
def test_wireframe_frames_equal_the_oracle(oracle, oracle_scene, mesh, W, H, frame):
    s, osc = R.Scene(R.assets.mesh_path(mesh)), oracle_scene(mesh)
    cam, lights, n = R.benchmark_frame(frame)
    ocam, olights, on = oracle.benchmark_frame(frame)
    g = s.render(3, cam, lights, n, R.default_opts(W, H))
    o = osc.render(3, ocam, olights, on, oracle.default_opts(W, H))
    assert (o[0] != 0).sum() > 20
    assert np.array_equal(g[0], o[0]), "%d pixels differ" % (g[0] != o[0]).sum()


def test_wireframe_cameras_inside_and_near_the_model(oracle, oracle_scene):
    """vertices behind the clip plane drop lines, projected coordinates wrap through 16 bits, lines are clipped"""
    s, osc = R.Scene(R.assets.mesh_path("chessboard.tri")), oracle_scene("chessboard.tri")
    rng = np.random.default_rng(23)
    _, lights, n = R.benchmark_frame(0)
    _, olights, on = oracle.benchmark_frame(0)
    for i in range(16):
        eye = rng.uniform(-1.0, 1.0, 3) * (0.3 if i % 2 else 1.5)
        at = rng.uniform(-0.5, 0.5, 3)
        if np.linalg.norm(np.cross(at - eye, [0, 0, 1])) < 1e-3:
            continue
        cam, ocam = R.camera(eye.astype(np.float32), at.astype(np.float32)), oracle.camera(eye.astype(np.float32), at.astype(np.float32))
        g = s.render(3, cam, lights, n, R.default_opts(480, 270))
        o = osc.render(3, ocam, olights, on, oracle.default_opts(480, 270))
        assert np.array_equal(g[0], o[0]), "camera %d: %d pixels differ" % (i, (g[0] != o[0]).sum())


def test_wireframe_bands_pitch_and_the_cli(oracle, oracle_scene, tmp_path):
    s, osc = R.Scene(R.assets.mesh_path("dragon_vis.ply")), oracle_scene("dragon_vis.ply")
    W, H = 640, 360
    cam, lights, n = R.benchmark_frame(12)
    ocam, olights, on = oracle.benchmark_frame(12)
    whole = osc.render(3, ocam, olights, on, oracle.default_opts(W, H))[0]
    out = np.zeros_like(whole)
    for b in range(3):
        part = s.render(3, cam, lights, n, R.default_opts(W, H, band_rows=8, band_index=b, band_count=3))[0]
        rows = [y for y in range(H) if (y // 8) % 3 == b]
        out[rows] = part[rows]
    assert np.array_equal(out, whole)
    wide = s.render(3, cam, lights, n, R.default_opts(W, H), pitch_words=W + 24)[0]
    assert np.array_equal(wide, whole)
    # the C++ host layer: Scene::renderWireframe through render_cli -m 3
    import subprocess
    r = subprocess.run([R.RENDER_CLI, "-b", "-n", "2", "-m", "3", "-W", str(W), "-H", str(H), "-o", str(tmp_path / "wf"), R.assets.mesh_path("dragon_vis.ply")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    ppm = (tmp_path / "wf_0001.ppm").read_bytes()
    rgb = np.frombuffer(ppm[-W * H * 3:], np.uint8).reshape(H, W, 3).astype(np.uint32)
    ocam0 = oracle.benchmark_frame(0)[0]          # (the CLI's first frame is frame 0 of the orbit)
    ref = osc.render(3, ocam0, olights, on, oracle.default_opts(W, H))[0]
    assert np.array_equal((rgb[..., 0] << 16) | (rgb[..., 1] << 8) | rgb[..., 2], ref)
