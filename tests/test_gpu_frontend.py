"""The interactive loop (frontend.h) WITH rendering: every frame it decides on is drawn on the GPU through the C++ host layer; the
canvas after a key script holds the frame the last camera / light / mode of the trace gives when rendered directly."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import renderer_amd as R

pytestmark = pytest.mark.gpu

SCRIPT = """
poll 4
down left
poll 6
up left
down w
poll 3
up w
tap 7
poll 5
tap 9        # frozen raytraced frame, released with ESC, then soft shadow maps again
poll 700
tap esc
poll 3
"""


def test_rendered_key_trace(oracle, oracle_scene):
    W, H = 320, 240
    mesh = "chessboard.tri"
    s = R.Scene(R.assets.mesh_path(mesh))
    s.bvh_create()
    f = R.host().mi355h_frontend_trace
    f.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long, C.c_void_p, C.c_int, C.c_void_p]
    f.restype = C.c_int
    rows = np.zeros((4000, 24), np.float32)
    last = np.zeros((H, W), np.uint32)
    n = f(s._h, SCRIPT.encode(), 6, 0, 1, W, H, 10, rows.ctypes.data, 4000, last.ctypes.data)
    assert n > 15, R.host().mi355h_last_error().decode()
    rows = rows[:n]
    modes = rows[:, 1].astype(int).tolist()
    assert 9 in modes and modes[-1] == 8 and 7 in modes
    # the last frame, rendered directly by the oracle from the trace's own camera and light: mode 8 after the light was moved in
    # mode 6 -- the front-end redrew the shadow map when mode 7 was chosen (renderer.cc:466-470), never since
    r = rows[-1]
    osc = oracle_scene(mesh)
    cam = oracle.camera(r[2:5], r[5:8])
    assert np.array_equal(np.array(list(cam.mv), np.float32).view(np.uint32), r[11:20].view(np.uint32))
    light = oracle.light(r[8:11], cam)
    lights = (oracle.Light * 2)(light)
    ref = osc.render(8, cam, lights, 1, oracle.default_opts(W, H), shadow_maps=[osc.shadowmap(light)])[0]
    assert np.array_equal(last, ref)


@pytest.mark.parametrize("mode_key,mode,H", [("9", 9, 240), ("0", 10, 100)])
def test_the_frozen_raytraced_frame_is_the_whole_frame(oracle, oracle_scene, mode_key, mode, H):
    """The raytraced modes draw their frame 16 scanlines at a time between the keyboard polls (Raytracer.cc:812-866 under
    HANDLERAYTRACER).  The finished frame -- what stays on the screen until ESC -- must be the frame a direct render gives:
    every band in its own rows, no band disturbing another (each band call used to black out the earlier ones).
    The script ends while the picture is frozen, so the canvas handed back IS that picture.  H = 100: a ragged last band."""
    W = 320
    mesh = "dragon_vis.ply"
    s = R.Scene(R.assets.mesh_path(mesh))
    s.bvh_create()
    f = R.host().mi355h_frontend_trace
    f.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long, C.c_void_p, C.c_int, C.c_void_p]
    f.restype = C.c_int
    rows = np.zeros((100, 24), np.float32)
    last = np.zeros((H, W), np.uint32)
    script = "poll 3\ntap %s\npoll %d\n" % (mode_key, H + 20)
    n = f(s._h, script.encode(), 6, 0, 1, W, H, 10, rows.ctypes.data, 100, last.ctypes.data)
    assert n >= 3, R.host().mi355h_last_error().decode()
    r = rows[n - 1]
    assert int(r[1]) == mode and r[22] == 1                               # the last frame drawn: raytraced, completed
    cam = R.camera(r[2:5], r[5:8])
    lt = R.light(r[8:11], cam)
    lights = (R.Light * 2)(lt)
    direct = s.render(mode, cam, lights, 1, R.default_opts(W, H))[0]
    assert int((direct != 0).sum()) > W * H // 50
    assert np.array_equal(last, direct)
    osc = oracle_scene(mesh, bvh=True)
    ocam = oracle.camera(r[2:5], r[5:8])
    ol = oracle.light(r[8:11], ocam)
    ref = osc.render(mode, ocam, (oracle.Light * 2)(ol), 1, oracle.default_opts(W, H, threads=os.cpu_count() or 1))[0]
    assert np.array_equal(last, ref)


def test_render_cli_keys(tmp_path):
    cli = os.path.join(os.path.dirname(R.RENDER_SO), "render_cli")
    keys = tmp_path / "keys.txt"
    keys.write_text("poll 3\ntap 4\npoll 3\ntap pgup\npoll 2\n")
    out = subprocess.run([cli, "--keys", str(keys), "--frame-ms", "10", "-m", "6", "-W", "160", "-H", "120", "-o", str(tmp_path / "k"), R.assets.mesh_path("trainColor.tri")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("frame ")]
    assert len(lines) >= 7 and " mode 4 " in out.stdout and " mode 5 " in out.stdout
    assert os.path.getsize(str(tmp_path / ("k_%04d.ppm" % len(lines)))) == 160 * 120 * 3 + len("P6\n160 120\n255\n")
