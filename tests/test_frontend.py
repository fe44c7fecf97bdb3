"""The interactive frame loop without a window system (renderer_amd/csrc/host/frontend.{h,cc}: the reference's
renderer.cc:243-642 + Keyboard.cc + the scanline polls of Raytracer.cc:840-864 as a state machine) against the oracle's
restatement of the same loop (oracle.cc: orc_frontend_trace): the sequence of frames a key script leads to -- pass of the loop,
mode, eye, lookat, light, camera matrix, rotation step -- bit for bit.  Dry runs: nothing is rendered, so no GPU is needed."""
import ctypes as C

import numpy as np
import pytest

import renderer_amd as R


def host_trace(script, mode=8, two=False, brakes=True, frame_ms=10, max_frames=4000, height=600):
    out = np.zeros((max_frames, 24), np.float32)
    f = R.host().mi355h_frontend_trace
    f.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long, C.c_void_p, C.c_int, C.c_void_p]
    f.restype = C.c_int
    n = f(None, script.encode(), mode, int(two), int(brakes), 800, height, frame_ms, out.ctypes.data, max_frames, None)
    assert n >= 0, R.host().mi355h_last_error().decode()
    return out[:min(n, max_frames)], n


def oracle_trace(oracle, script, mode=8, two=False, brakes=True, frame_ms=10, max_frames=4000, height=600):
    out = np.zeros((max_frames, 24), np.float32)
    f = oracle.lib().orc_frontend_trace
    f.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long, C.c_void_p, C.c_int]
    f.restype = C.c_int
    n = f(script.encode(), mode, int(two), int(brakes), height, frame_ms, out.ctypes.data, max_frames)
    assert n >= 0
    return out[:min(n, max_frames)], n


def same(a, b):
    return a.shape == b.shape and bool((a.view(np.uint32) == b.view(np.uint32)).all())


WALK = """
poll 20                 # the auto-spin orbit, untouched
down left
poll 15
down up
poll 10
up left
poll 12
up up
tap r                   # free look: angles from the eye's position
poll 5
down a
poll 8
up a
down s
poll 4
down f                  # S and F together: the right axis is scaled twice (renderer.cc:386-387)
poll 3
up s
up f
down e
poll 6
up e
down d
poll 2
up d
down z
poll 3
up z
down right
down down
poll 30
up right
up down
tap r                   # back to the orbit: angles negated
poll 25
down w                  # the light moves (mode 8: its shadow map is redrawn)
poll 9
up w
down q
poll 4
up q
tap pgup                # mode 9 ... wraps through the list
poll 2
tap 5
poll 7
tap pgdn
tap pgdn
poll 3
down h                  # help: H down, up, then H again to leave
up h
poll 3
tap h
poll 12
"""


@pytest.mark.parametrize("frame_ms", [10, 3, 40])
def test_a_key_trace_leads_to_the_same_frames(oracle, frame_ms):
    got, n = host_trace(WALK, mode=8, brakes=False, frame_ms=frame_ms)
    want, m = oracle_trace(oracle, WALK, mode=8, brakes=False, frame_ms=frame_ms)
    assert n == m and n >= 200
    assert same(got, want), "first differing frame: %d" % int(np.flatnonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=1))[0])
    modes = got[:, 1].astype(int)
    assert {4, 5, 8, 9}.issubset(set(modes.tolist()))                      # the mode keys took effect
    assert len(np.unique(got[:, 8])) > 5 and (got[:, 21] == 0).sum() > 20  # the light moved; part of the walk was free look
    assert len(np.unique(got[:, 20])) > 50                                 # the rotation step follows the frame rate


def test_the_orbit_of_the_untouched_loop_is_the_benchmark_orbit():
    """No key at all and a clock that never advances: the loop's auto-spin is the benchmark's orbit (renderer.cc:485-494)."""
    got, n = host_trace("poll 60\n", mode=6, brakes=False, frame_ms=0)
    assert n == 60            # (one poll before the loop, one per pass: the eye moves every pass; the script's end closes the window)
    for k in (0, 17, 59):
        cam, lights, _ = R.benchmark_frame(k)
        assert np.array_equal(got[k, 2:5].view(np.uint32), np.array(list(cam.eye), np.float32).view(np.uint32))
        assert np.array_equal(got[k, 11:20].view(np.uint32), np.array(list(cam.mv), np.float32).view(np.uint32))


def test_raytraced_modes_freeze_the_frame_and_can_be_aborted(oracle):
    """HANDLERAYTRACER (renderer.cc:553-573, Raytracer.cc:840-864): a raytraced frame polls the keyboard once per scanline; ESC
    during the frame abandons it, ESC after it releases the frozen picture; either way the loop goes on in mode 8."""
    script = """
poll 3
tap 9                   # raytrace: 600 polls while the frame is traced, then frozen until ESC
poll 700
tap esc
poll 4
tap 0                   # anti-aliased: aborted after 100 scanlines
poll 100
down esc
up esc
poll 6
"""
    got, n = host_trace(script, mode=6, brakes=True)
    want, m = oracle_trace(oracle, script, mode=6, brakes=True)
    assert n == m and same(got, want)
    rt = got[got[:, 1] >= 9]
    assert rt.shape[0] == 2 and rt[0, 22] == 1 and rt[1, 22] == 0          # the first completed, the second was abandoned
    after = got[np.flatnonzero(got[:, 1] >= 9)[0] + 1]
    assert after[1] == 8                                                   # back in soft shadow maps
    # without the brakes (configure --disable-brakes) the raytraced modes are modes like the others
    got2, n2 = host_trace("poll 3\ntap 9\npoll 5\n", mode=6, brakes=False)
    want2, m2 = oracle_trace(oracle, "poll 3\ntap 9\npoll 5\n", mode=6, brakes=False)
    assert n2 == m2 and same(got2, want2) and (got2[:, 1] == 9).sum() >= 5


def test_keys_typed_during_a_raytraced_frame_go_to_the_frames_own_keyboard(oracle):
    """Raytracer.cc:812: renderRaytracer polls a LOCAL Keyboard.  A key that goes DOWN while the frame is traced is consumed there
    and never reaches the loop's flags (no mode change, no rotation afterwards); a key that goes UP during the frame leaves the
    loop's flag set (the eye keeps moving until the key comes up again outside a frame)."""
    script = """
poll 3
down left               # held into the frame ...
poll 2
tap 9
poll 100
up left                 # ... and released while it is traced: the loop never sees the release
poll 50
tap 4                   # a mode key typed during the frame is swallowed
down w
poll 500
tap esc                 # the frozen frame is released
poll 6                  # (left is still down for the loop: angle1 keeps changing; W's release comes now, its press never arrived)
up w
poll 4
down left
up left                 # only now does the loop's flag clear
poll 5
"""
    got, n = host_trace(script, mode=6, brakes=True)
    want, m = oracle_trace(oracle, script, mode=6, brakes=True)
    assert n == m and same(got, want), "first differing frame: %d" % int(np.flatnonzero((got.view(np.uint32) != want.view(np.uint32)).any(axis=1))[0])
    i9 = int(np.flatnonzero(got[:, 1] == 9)[0])
    assert got[i9, 22] == 1                                                # completed: the ESC-less keys did not abort it
    after = got[i9 + 1:]
    assert (after[:, 1] == 8).all()                                        # "tap 4" never reached the loop
    assert len(np.unique(after[:, 8])) == 1                                # nor did W: the light stayed where it was
    # (left is still held for the loop after the frame: the orbit's own step plus the key's, twice the untouched loop's)
    plain, _ = host_trace("poll 3\ndown left\npoll 2\nup left\ntap 9\npoll 650\ntap esc\npoll 6\n", mode=6, brakes=True)
    j9 = int(np.flatnonzero(plain[:, 1] == 9)[0])
    assert not same(after[:4, 2:5].copy(), plain[j9 + 1:j9 + 5, 2:5].copy())


def test_a_script_line_that_cannot_be_read_is_an_error():
    f = R.host().mi355h_frontend_trace
    f.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_long, C.c_void_p, C.c_int, C.c_void_p]
    f.restype = C.c_int
    out = np.zeros((4, 24), np.float32)
    assert f(None, b"poll 3\npress x\n", 6, 0, 1, 800, 600, 10, out.ctypes.data, 4, None) < 0
    assert b"key script line 2" in R.host().mi355h_last_error()
