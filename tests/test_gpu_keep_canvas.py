"""mi355_opts::keep_canvas: raster frames of the synchronous seam written straight into a page-locked canvas whose last frame is
known, and only where they can differ from it (the 64x64-pixel bins that hold triangles now, black into those that held some
before; k_rs_tile, capi.hip).  Whatever happens between two such frames -- other modes, other sizes, other canvases, frames of
the other entry points, buffers released -- every frame must be the frame the plain call draws (Rasterizers.cc:320-356: the
reference clears and draws the whole canvas every frame)."""
import numpy as np
import pytest

import renderer_amd as R

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

GARBAGE = 0x00C0FFEE


@pytest.fixture(scope="module")
def scene():
    s = R.Scene(R.assets.mesh_path("chessboard.tri"))
    cam, lights, n = R.benchmark_frame(0)
    s.shadowmap_render(0, lights[0])
    return s


def plain(scene, mode, k, W, H):
    cam, lights, n = R.benchmark_frame(k)
    return scene.render(mode, cam, lights, n, R.default_opts(W, H))[0]


def kept(scene, mode, k, canvas, W, H, keep=1):
    cam, lights, n = R.benchmark_frame(k)
    scene.render_into(mode, cam, lights, n, R.default_opts(W, H, keep_canvas=keep), canvas)


@pytest.mark.parametrize("W,H", [(1920, 1080), (333, 217), (64, 64), (1, 1)])
def test_kept_canvas_holds_the_plain_frames(scene, W, H):
    """Frames along the orbit (neighbours, far jumps, back again) and every filler, into ONE canvas that starts as garbage."""
    canvas = R.host_array((H, W))
    try:
        canvas[:] = GARBAGE
        for i, k in enumerate((0, 1, 2, 40, 41, 120, 0, 199, 3)):
            mode = (6, 8, 4, 7, 5)[i % 5]
            kept(scene, mode, k, canvas, W, H)
            assert np.array_equal(canvas, plain(scene, mode, k, W, H)), "frame %d mode %d" % (k, mode)
    finally:
        R.host_array_free(canvas)


def test_only_the_bins_that_can_differ_are_written(scene):
    """The point of the option -- and why it is a promise: a word scribbled where neither the last frame nor this one has
    triangles stays (keep_canvas = 1), and goes with keep_canvas = 2 (content unknown: written in full) and with 0."""
    W, H = 1920, 1080
    canvas = R.host_array((H, W))
    try:
        kept(scene, 6, 10, canvas, W, H)
        ref = plain(scene, 6, 11, W, H)
        assert not ref[:64, :64].any() and not plain(scene, 6, 10, W, H)[:64, :64].any()      # (the corner bin is background in both)
        canvas[5, 7] = GARBAGE
        kept(scene, 6, 11, canvas, W, H, keep=1)
        assert canvas[5, 7] == GARBAGE
        canvas[5, 7] = 0
        assert np.array_equal(canvas, ref)
        canvas[5, 7] = GARBAGE
        kept(scene, 6, 12, canvas, W, H, keep=2)
        assert np.array_equal(canvas, plain(scene, 6, 12, W, H))
        canvas[5, 7] = GARBAGE
        kept(scene, 6, 13, canvas, W, H, keep=0)
        assert np.array_equal(canvas, plain(scene, 6, 13, W, H))
        # ... and a frame behind a plain one starts from scratch by itself
        canvas[5, 7] = GARBAGE
        kept(scene, 6, 14, canvas, W, H, keep=1)
        assert np.array_equal(canvas, plain(scene, 6, 14, W, H))
    finally:
        R.host_array_free(canvas)


def test_other_frames_in_between_make_the_canvas_unknown(scene):
    """Points, wireframe, a raytraced frame, a frame of the asynchronous entry point, a counting frame and a band into the same
    canvas: the kept frame behind each of them is complete."""
    W, H = 640, 360
    ray = R.Scene(R.assets.mesh_path("chessboard.tri"))
    ray.bvh_create()
    canvas = R.host_array((H, W))
    try:
        kept(scene, 6, 0, canvas, W, H)
        k = 1
        for what in ("points", "wire", "async", "counting", "band", "mlaa", "other context"):
            cam, lights, n = R.benchmark_frame(50 + k)
            if what == "points":
                scene.render_into(2, cam, lights, n, R.default_opts(W, H), canvas)
            elif what == "wire":
                scene.render_into(3, cam, lights, n, R.default_opts(W, H), canvas)
            elif what == "async":
                scene.render_wait(scene.render_async(8, cam, lights, n, R.default_opts(W, H), canvas))
            elif what == "counting":
                scene.render_into(6, cam, lights, n, R.default_opts(W, H, collect_stats=1, keep_canvas=1), canvas)
                assert np.array_equal(canvas, plain(scene, 6, 50 + k, W, H))
            elif what == "band":
                scene.render_into(6, cam, lights, n, R.default_opts(W, H, band_rows=8, band_index=1, band_count=2, keep_canvas=1), canvas)
            elif what == "mlaa":
                scene.render_into(6, cam, lights, n, R.default_opts(W, H, mlaa=1, keep_canvas=1), canvas)
            else:
                ray.render_into(9, cam, lights, n, R.default_opts(W, H), canvas)
            for j in range(2):
                kept(scene, (6, 8)[j], k, canvas, W, H)
                assert np.array_equal(canvas, plain(scene, (6, 8)[j], k, W, H)), "behind %s, frame %d" % (what, k)
                k += 1
    finally:
        R.host_array_free(canvas)


def test_canvases_sizes_and_pitches_take_turns(scene):
    """Two canvases of one size, one of another, one with padding words (which are never touched), each with the frames it
    held before."""
    a, b = R.host_array((360, 640)), R.host_array((360, 640))
    c, d = R.host_array((217, 333)), R.host_array((360, 640 + 16))
    try:
        d[:] = GARBAGE
        for i in range(12):
            canvas = (a, b, a, a, c, a, d, d, b, d, c, c)[i]
            H, W = (360, 640) if canvas is not c else (217, 333)
            kept(scene, 6, 3 * i, canvas, W, H)
            assert np.array_equal(canvas[:, :W], plain(scene, 6, 3 * i, W, H)), "turn %d" % i
        assert (d[:, 640:] == GARBAGE).all()
    finally:
        for x in (a, b, c, d):
            R.host_array_free(x)


def test_a_released_canvas_is_forgotten(scene):
    """A canvas is freed and the next allocation may lie at the same address, zero-filled: not the frame the masks describe."""
    W, H = 640, 360
    for i in range(4):
        canvas = R.host_array((H, W))
        try:
            canvas[:] = GARBAGE
            kept(scene, 6, 20 * i, canvas, W, H)
            assert np.array_equal(canvas, plain(scene, 6, 20 * i, W, H))
        finally:
            R.host_array_free(canvas)
    # ... and caller's memory that is registered, drawn into, unregistered
    mem = R.own_mapping_array((H, W))
    scene.host_register(mem)
    try:
        for k in (0, 5):
            mem[3, 3] = GARBAGE
            kept(scene, 8, k, mem, W, H, keep=2 if k == 0 else 1)
            assert mem[3, 3] == (0 if k == 0 else GARBAGE)
            mem[3, 3] = 0
            assert np.array_equal(mem, plain(scene, 8, k, W, H))
    finally:
        scene.host_unregister(mem)
    # pageable memory: the option does not apply, the frame is complete
    page = np.full((H, W), GARBAGE, np.uint32)
    kept(scene, 6, 7, page, W, H)
    assert np.array_equal(page, plain(scene, 6, 7, W, H))


def _write_soup(path, verts, tris, cols):
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n" % len(verts))
        f.write("element face %d\nproperty list uchar int vertex_indices\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % len(tris))
        for v in verts:
            f.write("%r %r %r\n" % (float(v[0]), float(v[1]), float(v[2])))
        for t, c in zip(tris, cols):
            f.write("3 %d %d %d %d %d %d\n" % (t[0], t[1], t[2], c[0], c[1], c[2]))


def test_triangles_in_the_global_bin_and_a_frame_without_any(tmp_path):
    """Seven triangles larger than the view (every tile holds them: nothing of the background is anybody's to write), then the
    camera turned away (no triangle at all: every bin of the frame before is blackened), then back."""
    rng = np.random.default_rng(11)
    v = rng.uniform(-1, 1, (7, 1, 3)) + rng.uniform(-2.5, 2.5, (7, 3, 3))
    p = str(tmp_path / "huge.ply")
    _write_soup(p, v.reshape(-1, 3), np.arange(21).reshape(7, 3), rng.integers(30, 255, (7, 3)))
    s = R.Scene(p)
    W, H = 333, 217
    eye = np.array([-0.6, 0.9, 0.4], np.float32)
    inside, away = np.array([0.1, 0.0, -0.1], np.float32), np.array([-60.0, 90.0, 40.0], np.float32)
    lp = np.array([2.0, -1.0, 2.5], np.float32)
    canvas = R.host_array((H, W))
    try:
        canvas[:] = GARBAGE
        for i, look in enumerate((inside, away, inside, inside, away, away)):
            cam = R.camera(eye, look)
            lights = (R.Light * 2)(R.light(lp, cam))
            s.render_into(6, cam, lights, 1, R.default_opts(W, H, keep_canvas=1), canvas)
            want = s.render(6, cam, lights, 1, R.default_opts(W, H))[0]
            assert np.array_equal(canvas, want), "frame %d" % i
    finally:
        R.host_array_free(canvas)


def test_bins_that_overflow_are_drawn_again_into_the_canvas(tmp_path):
    """mi355_render draws a frame again when the rasterizer's buffers had to grow: the second pass takes the same masks."""
    rng = np.random.default_rng(5)
    n = 150000
    c = rng.uniform(-0.2, 0.2, (n, 1, 3))
    v = c + rng.uniform(-1.0, 1.0, (n, 3, 3)) * np.array([0.05, 1.0, 1.0])
    p = str(tmp_path / "layers.ply")
    _write_soup(p, v.reshape(-1, 3), np.arange(3 * n).reshape(n, 3), rng.integers(30, 255, (n, 3)))
    W, H = 200, 150
    eye, look = np.array([2.2, 0.2, 0.1], np.float32), np.array([0.0, 0.0, 0.0], np.float32)
    cam = R.camera(eye, look)
    lights = (R.Light * 2)(R.light(np.array([3.0, 1.0, 1.0], np.float32), cam))
    want = R.Scene(p).render(6, cam, lights, 1, R.default_opts(W, H))[0]
    s = R.Scene(p)                         # (a fresh context: its buffers have not grown yet)
    canvas = R.host_array((H, W))
    try:
        canvas[:] = GARBAGE
        s.render_into(6, cam, lights, 1, R.default_opts(W, H, keep_canvas=1), canvas)
        assert np.array_equal(canvas, want)
        s.render_into(6, cam, lights, 1, R.default_opts(W, H, keep_canvas=1), canvas)
        assert np.array_equal(canvas, want)
    finally:
        R.host_array_free(canvas)


def test_frames_in_flight_keep_their_canvases(scene):
    """mi355_render_async: a ring of three canvases, three frames in flight, every frame written where it differs from the frame
    that canvas held three frames ago; then the ring shrinks to ONE canvas used by every slot in turn (its frames follow each
    other across the slots' streams), a synchronous kept frame and a plain one in between."""
    W, H = 1280, 720
    ring = [R.host_array((H, W)) for _ in range(3)]
    try:
        for c in ring:
            c[:] = GARBAGE
        o = R.default_opts(W, H, keep_canvas=1)
        tickets = [None, None, None]
        frame_of = [0, 0, 0]
        checked = 0
        for f in range(30):
            slot = f % 3
            if tickets[slot] is not None:
                scene.render_wait(tickets[slot])
                assert np.array_equal(ring[slot], plain(scene, frame_of[slot][0], frame_of[slot][1], W, H)), "frame %d" % f
                checked += 1
            mode, k = (6, 8, 4)[(f // 3) % 3], 5 * f
            cam, lights, n = R.benchmark_frame(k)
            tickets[slot] = scene.render_async(mode, cam, lights, n, o, ring[slot])
            frame_of[slot] = (mode, k)
        for slot in range(3):
            scene.render_wait(tickets[slot])
            assert np.array_equal(ring[slot], plain(scene, frame_of[slot][0], frame_of[slot][1], W, H))
        assert checked == 27
        one = ring[0]
        for f in range(12):
            cam, lights, n = R.benchmark_frame(7 * f)
            if f % 4 == 2:
                scene.render_into(6, cam, lights, n, o, one)                           # the synchronous call, kept
            elif f % 4 == 3:
                scene.render_into(6, cam, lights, n, R.default_opts(W, H), one)        # ... and plain
            else:
                scene.render_wait(scene.render_async(6, cam, lights, n, o, one))
            assert np.array_equal(one, plain(scene, 6, 7 * f, W, H)), "one canvas, frame %d" % f
        # five canvases take turns: one more than the context remembers
        more = ring + [R.host_array((H, W)), R.host_array((H, W))]
        ring = more
        for f in range(15):
            cam, lights, n = R.benchmark_frame(3 * f)
            c = more[f % 5]
            scene.render_wait(scene.render_async(8, cam, lights, n, o, c))
            assert np.array_equal(c, plain(scene, 8, 3 * f, W, H)), "five canvases, frame %d" % f
    finally:
        for c in ring:
            R.host_array_free(c)


@pytest.fixture(scope="module")
def ray_scene():
    s = R.Scene(R.assets.mesh_path("dragon_vis.ply"))
    s.bvh_create()
    cam, lights, n = R.benchmark_frame(0)
    s.shadowmap_render(0, lights[0])
    return s


@pytest.mark.parametrize("W,H", [(1920, 1080), (333, 217), (8, 8)])
def test_raytraced_frames_keep_their_canvas(ray_scene, W, H):
    """Modes 9 and 10 by 8x8 tile: the tiles a frame traces are written by their waves, black goes into those the canvas's last
    frame traced; a rasterized frame in between changes what the masks mean (the frame behind it is written in full)."""
    s = ray_scene
    canvas = R.host_array((H, W))
    try:
        canvas[:] = GARBAGE
        for i, k in enumerate((0, 1, 2, 60, 61, 150, 0, 3, 4, 5)):
            mode = (9, 9, 9, 9, 10, 9, 6, 9, 9, 8)[i] if W < 1000 else (9, 9, 9, 9, 9, 9, 6, 9, 9, 9)[i]
            kept(s, mode, k, canvas, W, H)
            assert np.array_equal(canvas, plain(s, mode, k, W, H)), "frame %d mode %d" % (k, mode)
        if W == 1920:
            # (the promise: a corner tile neither frame traces is not written)
            canvas[2, 3] = GARBAGE
            kept(s, 9, 6, canvas, W, H)
            assert canvas[2, 3] == GARBAGE
            kept(s, 9, 7, canvas, W, H, keep=2)
            assert np.array_equal(canvas, plain(s, 9, 7, W, H))
            # (the exact box test only, tune flag 1: the tiles are culled all the same, the canvas is kept)
            cam, lights, n = R.benchmark_frame(8)
            s.render_into(9, cam, lights, n, R.default_opts(W, H, keep_canvas=1, tune=R.tune(exact=1)), canvas)
            assert np.array_equal(canvas, plain(s, 9, 8, W, H))
            # three in flight, a canvas each
            ring = [canvas, R.host_array((H, W)), R.host_array((H, W))]
            try:
                o = R.default_opts(W, H, keep_canvas=1)
                tickets, frame_of = [None] * 3, [0] * 3
                for f in range(12):
                    slot = f % 3
                    if tickets[slot] is not None:
                        s.render_wait(tickets[slot])
                        assert np.array_equal(ring[slot], plain(s, 9, frame_of[slot], W, H)), "in flight, frame %d" % f
                    cam, lights, n = R.benchmark_frame(11 * f)
                    tickets[slot] = s.render_async(9, cam, lights, n, o, ring[slot])
                    frame_of[slot] = 11 * f
                for slot in range(3):
                    s.render_wait(tickets[slot])
                    assert np.array_equal(ring[slot], plain(s, 9, frame_of[slot], W, H))
            finally:
                for c in ring[1:]:
                    R.host_array_free(c)
    finally:
        R.host_array_free(canvas)


def test_raytraced_keep_is_ignored_where_tiles_are_not_culled(ray_scene):
    """Counting frames, float output, the reference-order walk and every-tile frames (tune flags 4, 16): no tile mask, no kept
    canvas -- complete frames all the same."""
    s = ray_scene
    W, H = 320, 240
    canvas = R.host_array((H, W))
    try:
        kept(s, 9, 0, canvas, W, H)
        for i, kw in enumerate((dict(tune=R.tune(reforder=1)), dict(tune=R.tune(nocull=1)), dict(collect_stats=1))):
            cam, lights, n = R.benchmark_frame(10 + i)
            canvas[1, 1] = GARBAGE
            s.render_into(9, cam, lights, n, R.default_opts(W, H, keep_canvas=1, **kw), canvas)
            assert np.array_equal(canvas, plain(s, 9, 10 + i, W, H)), str(kw)
            kept(s, 9, 20 + i, canvas, W, H)
            assert np.array_equal(canvas, plain(s, 9, 20 + i, W, H)), "behind " + str(kw)
    finally:
        R.host_array_free(canvas)


def test_cxx_screen_keeps_its_canvas():
    """The host layer: Screen::_keepCanvas through Scene::renderPhong, ClearScreen() and touched() in between (render_cli
    --keep-canvas dumps the frames it presents)."""
    import hashlib, os, subprocess, tempfile
    cli = os.path.join(os.path.dirname(R.RENDER_SO), "render_cli")
    mesh = R.assets.mesh_path("chessboard.tri")
    with tempfile.TemporaryDirectory() as d:
        outs = []
        for m in ("6", "9"):
            outs = []
            for i, flag in enumerate(([], ["--keep-canvas"], ["--keep-canvas", "-p", "3"])):
                prefix = os.path.join(d, "run%s%d" % (m, i))
                subprocess.run([cli, "-b", "-n", "8", "-m", m, "-W", "640", "-H", "360", "-o", prefix] + (flag if "-p" in flag else flag + ["-p", "1"]) + [mesh],
                               check=True, capture_output=True)
                outs.append([hashlib.sha256(open("%s_%04d.ppm" % (prefix, f), "rb").read()).hexdigest() for f in range(1, 9)])
            assert outs[0] == outs[1] == outs[2], "mode " + m
