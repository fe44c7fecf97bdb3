"""Pipelined frames (mi355_render_async / mi355_render_wait) and caller-registered output buffers: the frames are the
frames mi355_render produces, whatever is in flight beside them."""
import numpy as np
import pytest

import renderer_amd as R

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode,mesh", [(9, "dragon_vis.ply"), (6, "chessboard.tri"), (8, "chessboard.tri"), (2, "chessboard.tri")])
@pytest.mark.parametrize("registered", [False, True, "allocated"])     # pageable / mi355_host_register / mi355_host_alloc
def test_pipelined_frames_equal_synchronous_frames(mode, mesh, registered):
    W, H = 640, 360
    s = R.Scene(R.assets.mesh_path(mesh))
    if mode >= 9:
        s.bvh_create()
    cams = [R.benchmark_frame(k) for k in range(0, 40, 4)]
    if mode in (7, 8):
        s.shadowmap_render(0, cams[0][1][0])
    o = R.default_opts(W, H)
    want = [s.render(mode, *c, o)[0] for c in cams]
    # (pitch wider than the frame; registered memory has a mapping of its own, not a piece of the heap: mi355_render.h)
    make = {False: lambda: np.empty((H, W + 8), np.uint32), True: lambda: R.own_mapping_array((H, W + 8)), "allocated": lambda: R.host_array((H, W + 8))}[registered]
    bufs = [make() for _ in range(R.MAX_IN_FLIGHT)]
    for b in bufs:
        b[:] = 0xdeadbeef
    if registered is True:
        for b in bufs:
            s.host_register(b)
    try:
        tickets = []
        got = []
        for k, c in enumerate(cams):
            if len(tickets) == R.MAX_IN_FLIGHT:
                t, slot = tickets.pop(0)
                st = s.render_wait(t)
                assert mode < 9 or st.normal_rays > 0
                got.append(bufs[slot][:, :W].copy())
                assert (bufs[slot][:, W:] == 0xdeadbeef).all()
            slot = k % R.MAX_IN_FLIGHT
            tickets.append((s.render_async(mode, *c, o, bufs[slot]), slot))
        with pytest.raises(R.Mi355Error, match="in flight"):
            s.render_async(mode, *cams[0], o, np.zeros((H, W), np.uint32))
        for t, slot in tickets:
            s.render_wait(t)
            got.append(bufs[slot][:, :W].copy())
        with pytest.raises(R.Mi355Error, match="no frame with ticket"):
            s.render_wait(12345)
    finally:
        if registered is True:
            for b in bufs:
                s.host_unregister(b)
        if registered == "allocated":
            for b in bufs:
                R.host_array_free(b)
    assert len(got) == len(want)
    for k in range(len(want)):
        assert np.array_equal(got[k], want[k]), "frame %d" % k


def test_synchronous_render_into_a_registered_buffer():
    W, H = 800, 600
    s = R.Scene(R.assets.mesh_path("dragon_vis.ply"))
    s.bvh_create()
    cam, lights, n = R.benchmark_frame(3)
    o = R.default_opts(W, H)
    want = s.render(9, cam, lights, n, o)[0]
    buf = R.own_mapping_array((H, W))
    s.host_register(buf)
    try:
        s.render_into(9, cam, lights, n, o, buf)
    finally:
        s.host_unregister(buf)
    assert np.array_equal(buf, want)
    with pytest.raises(R.Mi355Error, match="was not registered"):
        s.host_unregister(buf)


def test_synchronous_render_into_frame_memory_of_the_librarys():
    """mi355_host_alloc: the canvas the C++ host layer's Screen uses -- raytraced frames are written there by the kernels themselves
    (no copy), raster frames by one DMA transfer; several contexts draw into the same memory."""
    W, H = 800, 600
    buf = R.host_array((H, W))
    try:
        assert not buf.any()
        for mesh, mode in (("dragon_vis.ply", 9), ("chessboard.tri", 6), ("dragon_vis.ply", 10)):
            s = R.Scene(R.assets.mesh_path(mesh))
            if mode >= 9:
                s.bvh_create()
            for k in (3, 77):
                cam, lights, n = R.benchmark_frame(k)
                o = R.default_opts(W, H)
                want = s.render(mode, cam, lights, n, o)[0]
                buf[:] = 0x00c0ffee
                s.render_into(mode, cam, lights, n, o, buf)
                assert np.array_equal(buf, want), (mesh, mode, k)
    finally:
        R.host_array_free(buf)


def test_many_frame_geometries_in_one_context():
    """More frame sizes than the dispenser-order cache holds, revisited: every frame still equals a fresh context's."""
    s = R.Scene(R.assets.mesh_path("dragon_vis.ply"))
    s.bvh_create()
    cam, lights, n = R.benchmark_frame(1)
    sizes = [(64, 48), (320, 200), (333, 217), (640, 360), (800, 600), (64, 48), (1024, 768), (320, 200)]
    first = {}
    for (W, H) in sizes:
        img = s.render(9, cam, lights, n, R.default_opts(W, H))[0]
        if (W, H) in first:
            assert np.array_equal(img, first[(W, H)])
        else:
            first[(W, H)] = img
            fresh = R.Scene(R.assets.mesh_path("dragon_vis.ply"))
            fresh.bvh_create()
            assert np.array_equal(img, fresh.render(9, cam, lights, n, R.default_opts(W, H))[0])


@pytest.mark.parametrize("seed", range(6))
def test_frames_written_in_place_at_odd_sizes_pitches_and_cameras(seed):
    """The zero-copy frame (mi355_render into frame memory of the library's: traced tiles stored by their waves, the background by waves
    that start with it or have run out of pixels) where its bookkeeping has edges: widths and heights that are no multiple of the 8x8
    tiles, a pitch wider than the frame (the padding must stay untouched), a window into a larger canvas, cameras that see everything,
    a sliver or nothing of the model, 4 spp; every frame equals the frame the plain call returns."""
    rng = np.random.default_rng(900 + seed)
    mesh = ["dragon_vis.ply", "statue.ply", "chessboard.tri"][seed % 3]
    s = R.Scene(R.assets.mesh_path(mesh))
    s.bvh_create()
    canvas = R.host_array((1100, 2000))
    try:
        for trial in range(5):
            W, H = int(rng.integers(1, 1900)), int(rng.integers(1, 1000))
            x0, y0 = int(rng.integers(0, 2000 - W)), int(rng.integers(0, 1100 - H))
            kind = int(rng.integers(0, 4))
            if kind == 0:
                cam, lights, n = R.benchmark_frame(int(rng.integers(0, 200)))
            else:
                eye = (rng.normal(size=3) * [0.3, 1.5, 4.0][kind - 1]).astype(np.float32)
                at = (rng.normal(size=3) * 0.5).astype(np.float32) if kind != 3 else (eye * 2).astype(np.float32)
                if np.linalg.norm(np.cross(at - eye, [0, 0, 1])) < 1e-3:
                    continue
                cam = R.camera(eye, at)
                lights = (R.Light * 2)(R.light(np.array([3.0, 3.0, 4.0], np.float32), cam))
                n = 1
            mode = 10 if trial == 4 else 9
            o = R.default_opts(W, H)
            want = s.render(mode, cam, lights, n, o)[0]
            canvas[:] = 0x00abcdef
            view = canvas[y0:y0 + H, x0:x0 + W]
            s.render_into(mode, cam, lights, n, o, view)
            assert np.array_equal(view, want), "seed %d trial %d: %dx%d at (%d, %d), camera kind %d" % (seed, trial, W, H, x0, y0, kind)
            canvas[y0:y0 + H, x0:x0 + W] = 0x00abcdef
            assert (canvas == 0x00abcdef).all(), "seed %d trial %d: pixels outside the window were written" % (seed, trial)
    finally:
        R.host_array_free(canvas)


def test_host_array_free_takes_views_and_waits_only_for_frames_into_that_buffer():
    """host_array_free finds the allocation by address: a slice, a reshape or another dtype of the array releases it; an array that
    is not frame memory (or one that was released already) is a ValueError.  Releasing one buffer while a pipelined frame is on its
    way into ANOTHER one waits for nothing that is not its own (mi355_host_free looks at the frames in flight, not at whole devices):
    the other frame is complete and correct afterwards."""
    W, H = 320, 200
    a = R.host_array((H, W))
    view = a[7:, 5:].view(np.int32)
    R.host_array_free(view)
    with pytest.raises(ValueError):
        R.host_array_free(a)                       # released already
    with pytest.raises(ValueError):
        R.host_array_free(np.zeros((4, 4), np.uint32))
    s = R.Scene(R.assets.mesh_path("dragon_vis.ply"))
    s.bvh_create()
    cam, lights, n = R.benchmark_frame(11)
    o = R.default_opts(W, H)
    want = s.render(9, cam, lights, n, o)[0]
    keep, spare = R.host_array((H, W)), R.host_array((H, W))
    try:
        for _ in range(5):
            t = s.render_async(9, cam, lights, n, o, keep)
            R.host_array_free(spare)               # (a frame is in flight -- into `keep`)
            spare = R.host_array((H, W))
            s.render_wait(t)
            assert np.array_equal(keep, want)
            keep[:] = 0
        # ... and releasing the target itself while its frame is in flight waits for that frame (no write after free)
        t = s.render_async(9, cam, lights, n, o, keep)
        R.host_array_free(keep)
        keep = None
        s.render_wait(t)
    finally:
        if keep is not None:
            R.host_array_free(keep)
        R.host_array_free(spare)
