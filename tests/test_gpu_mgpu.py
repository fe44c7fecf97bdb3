"""One frame on several GPUs from one host thread (mi355_mgpu_*, mi355::Scene::_devices): interleaved 8-scanline bands,
one exchange, assembly on the first device.  A one-GPU box plays every rank by listing its device several times (transport
"copy": peer copies instead of the grouped RCCL send/recv, which refuses duplicate devices); the RCCL collective itself is
covered by tests/test_gpu_rccl.py."""
import ctypes as C
import subprocess

import numpy as np
import pytest

import renderer_amd as R

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode,mesh,ranks", [(9, "dragon_vis.ply", 2), (9, "dragon_vis.ply", 3), (9, "dragon_vis.ply", 8), (10, "dragon_vis.ply", 4),
                                             (6, "chessboard.tri", 2), (8, "chessboard.tri", 5), (2, "chessboard.tri", 4)])
def test_cxx_scene_on_several_ranks_equals_one_device(mode, mesh, ranks):
    W, H = 642, 363                                    # 46 bands: ranks own different numbers of rows, the last band is short
    eye, look, lp = [4.8, -0.5, 0.4], [0.0, 0.0, 0.0], [[3.4, 3.4, 4.8]]
    one = R.Scene(R.assets.mesh_path(mesh))
    if mode >= 9:
        one.bvh_create()
    want, st1 = one.render_frame_cxx(mode, W, H, eye, look, lp)
    many = R.Scene(R.assets.mesh_path(mesh))
    many.set_devices([0] * ranks)
    if mode >= 9:
        many.bvh_create()
    got, st = many.render_frame_cxx(mode, W, H, eye, look, lp)
    assert np.array_equal(got, want)
    if mode >= 9:
        assert (st.normal_rays, st.shadow_rays) == (st1.normal_rays, st1.shadow_rays)
    got2, _ = many.render_frame_cxx(mode, 320, 200, eye, look, lp)          # another geometry on the same set
    assert np.array_equal(got2, one.render_frame_cxx(mode, 320, 200, eye, look, lp)[0])


def test_mgpu_c_entry_points():
    L = R.lib()
    s = R.Scene(R.assets.mesh_path("dragon_vis.ply"))
    s.bvh_create()
    nodes, idx = s.bvh_arrays()
    L.mi355_mgpu_create.restype = C.c_void_p
    L.mi355_mgpu_create.argtypes = [C.POINTER(R.SceneDesc), C.POINTER(C.c_int), C.c_int]
    L.mi355_mgpu_transport.restype = C.c_char_p
    L.mi355_mgpu_transport.argtypes = [C.c_void_p]
    L.mi355_mgpu_destroy.argtypes = [C.c_void_p]
    L.mi355_mgpu_set_bvh.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    L.mi355_mgpu_render.argtypes = [C.c_void_p, C.c_int, C.POINTER(R.Camera), C.POINTER(R.Light), C.c_int, C.POINTER(R.Opts), C.c_void_p,
                                    C.c_int, C.c_void_p, C.POINTER(R.Stats)]
    cam, lights, n = R.benchmark_frame(5)
    o = R.default_opts(800, 600)
    want = s.render(9, cam, lights, n, o)[0]
    for devs, transport in (([0], b"none"), ([0, 0, 0, 0], b"copy")):
        m = L.mi355_mgpu_create(C.byref(s.desc), (C.c_int * len(devs))(*devs), len(devs))
        assert m, L.mi355_last_error()
        try:
            assert L.mi355_mgpu_transport(m) == transport
            assert L.mi355_mgpu_set_bvh(m, nodes.ctypes.data, nodes.shape[0], idx.ctypes.data, idx.shape[0]) == 0
            out = np.zeros((600, 800), np.uint32)
            st = R.Stats()
            assert L.mi355_mgpu_render(m, 9, C.byref(cam), lights, n, C.byref(o), out.ctypes.data, 3200, None, C.byref(st)) == 0, L.mi355_last_error()
            assert np.array_equal(out, want)
            bad = R.default_opts(800, 600, band_count=2, band_index=0, band_rows=8)
            assert L.mi355_mgpu_render(m, 9, C.byref(cam), lights, n, C.byref(bad), out.ctypes.data, 3200, None, None) == -20
        finally:
            L.mi355_mgpu_destroy(m)


def _mgpu_api():
    L = R.lib()
    L.mi355_mgpu_create.restype = C.c_void_p
    L.mi355_mgpu_create.argtypes = [C.POINTER(R.SceneDesc), C.POINTER(C.c_int), C.c_int]
    L.mi355_mgpu_destroy.argtypes = [C.c_void_p]
    L.mi355_mgpu_set_bvh.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
    L.mi355_mgpu_shadowmap_render.argtypes = [C.c_void_p, C.c_int, C.POINTER(R.Light), C.c_int, C.c_void_p]
    L.mi355_mgpu_render_batch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(R.Camera), C.POINTER(R.Light), C.c_int, C.POINTER(R.Opts),
                                          C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
    L.mi355_mgpu_wait.argtypes = [C.c_void_p, C.c_int, C.POINTER(R.Stats)]
    return L


@pytest.mark.parametrize("mode,mesh,ranks,frames", [(9, "dragon_vis.ply", 4, 4), (9, "dragon_vis.ply", 3, 1), (6, "chessboard.tri", 2, 3), (8, "chessboard.tri", 8, 2)])
def test_mgpu_steps_in_flight_equal_single_device_frames(mode, mesh, ranks, frames):
    """mi355_mgpu_render_batch / _wait: steps of several frames, two in flight (the third call has to wait for the first),
    every frame the frame one device renders; ray counts of a step summed over ranks and frames."""
    torch = pytest.importorskip("torch")
    L = _mgpu_api()
    W, H = 642, 363
    s = R.Scene(R.assets.mesh_path(mesh))
    if mode >= 9:
        s.bvh_create()
    m = L.mi355_mgpu_create(C.byref(s.desc), (C.c_int * ranks)(*([0] * ranks)), ranks)
    assert m, L.mi355_last_error()
    try:
        if mode >= 9:
            nodes, idx = s.bvh_arrays()
            assert L.mi355_mgpu_set_bvh(m, nodes.ctypes.data, nodes.shape[0], idx.ctypes.data, idx.shape[0]) == 0
        o = R.default_opts(W, H)
        steps = 3
        bufs = [[torch.zeros((H, W), dtype=torch.int32, device="cuda:0") for _ in range(frames)] for _ in range(steps)]
        tickets, want, rays = [], [], []
        for k in range(steps):
            fs = [k * frames + j for j in range(frames)]
            cams = (R.Camera * frames)(*[R.benchmark_frame(f)[0] for f in fs])
            lights = (R.Light * frames)(*[R.benchmark_frame(f)[1][0] for f in fs])
            if mode in (7, 8):           # (one light position for the whole orbit: its map is drawn once)
                s.shadowmap_render(0, lights[0])
                assert L.mi355_mgpu_shadowmap_render(m, 0, C.byref(lights[0]), 1024, None) == 0, L.mi355_last_error()
            outs = (C.c_void_p * frames)(*[b.data_ptr() for b in bufs[k]])
            t = C.c_int(0)
            if k >= 2:
                assert L.mi355_mgpu_render_batch(m, mode, frames, cams, lights, 1, C.byref(o), outs, W * 4, C.byref(t)) == -45
                st = R.Stats()
                assert L.mi355_mgpu_wait(m, tickets[k - 2], C.byref(st)) == 0, L.mi355_last_error()
                rays.append((st.normal_rays, st.shadow_rays))
            assert L.mi355_mgpu_render_batch(m, mode, frames, cams, lights, 1, C.byref(o), outs, W * 4, C.byref(t)) == 0, L.mi355_last_error()
            tickets.append(t.value)
            fr = [s.render(mode, R.benchmark_frame(f)[0], R.benchmark_frame(f)[1], 1, o) for f in fs]
            want.append([x[0] for x in fr])
            if k == 0:
                rays_want = (sum(x[2].normal_rays for x in fr), sum(x[2].shadow_rays for x in fr))
        for k in range(1, steps):
            assert L.mi355_mgpu_wait(m, tickets[k], None) == 0, L.mi355_last_error()
        assert L.mi355_mgpu_wait(m, tickets[0], None) == -45          # (waited for already)
        for k in range(steps):
            for j in range(frames):
                assert np.array_equal(bufs[k][j].cpu().numpy().view(np.uint32), want[k][j]), (k, j)
        if mode >= 9:
            assert rays[0] == rays_want
    finally:
        L.mi355_mgpu_destroy(m)


def test_a_bin_overflow_fails_every_raster_step_in_flight(oracle, tmp_path):
    """a device's bin-overflow word is sticky and one per context.  With two raster steps in flight the wait of the
    first used to report (and clear) an overflow of either, and the wait of the second then returned 0 for frames that may
    have dropped entries.  Now BOTH waits return -44; drawn again (the buffers have grown) the frames are the oracle's.
    Scene: 150 000 triangles that each cover most of a small view (tests/test_gpu_parity.py)."""
    torch = pytest.importorskip("torch")

    def _write_soup(path, verts, faces, cols):
        with open(path, "w") as f:
            f.write("ply\nformat ascii 1.0\nelement vertex %d\nelement face %d\nend_header\n" % (len(verts), len(faces)))
            for q in verts:
                f.write("%r %r %r 200\n" % (float(q[0]), float(q[1]), float(q[2])))
            for t, c in zip(faces, cols):
                f.write("3 %d %d %d %d %d %d\n" % (t[0], t[1], t[2], c[0], c[1], c[2]))

    L = _mgpu_api()
    rng = np.random.default_rng(5)
    n = 150000
    c = rng.uniform(-0.2, 0.2, (n, 1, 3))
    v = c + rng.uniform(-1.0, 1.0, (n, 3, 3)) * np.array([0.05, 1.0, 1.0])
    p = str(tmp_path / "layers.ply")
    _write_soup(p, v.reshape(-1, 3), np.arange(3 * n).reshape(n, 3), rng.integers(30, 255, (n, 3)))
    s, o_s = R.Scene(p), oracle.Scene(p)
    W, H, ranks = 200, 150, 2
    eyes = [np.array([2.2, 0.2, 0.1], np.float32), np.array([2.1, 0.5, 0.2], np.float32)]
    look = np.zeros(3, np.float32)
    lp = np.array([3.0, 1.0, 1.0], np.float32)
    m = L.mi355_mgpu_create(C.byref(s.desc), (C.c_int * ranks)(*([0] * ranks)), ranks)
    assert m, L.mi355_last_error()
    try:
        o = R.default_opts(W, H)
        bufs = [torch.zeros((H, W), dtype=torch.int32, device="cuda:0") for _ in range(2)]

        def enqueue(k):
            cam = R.camera(eyes[k], look)
            cams, lights = (R.Camera * 1)(cam), (R.Light * 1)(R.light(lp, cam))
            t = C.c_int(0)
            # (the destination array is a temporary on purpose: the library keeps its own copy)
            assert L.mi355_mgpu_render_batch(m, 4, 1, cams, lights, 1, C.byref(o), (C.c_void_p * 1)(bufs[k].data_ptr()), W * 4, C.byref(t)) == 0, L.mi355_last_error()
            return t.value

        t0, t1 = enqueue(0), enqueue(1)
        assert L.mi355_mgpu_wait(m, t0, None) == -44
        assert L.mi355_mgpu_wait(m, t1, None) == -44, "the second step in flight must be drawn again too"
        for attempt in range(8):          # (the buffers double per report)
            t0, t1 = enqueue(0), enqueue(1)
            r0, r1 = L.mi355_mgpu_wait(m, t0, None), L.mi355_mgpu_wait(m, t1, None)
            assert r0 in (0, -44) and r1 in (0, -44), L.mi355_last_error()
            if r0 == 0 and r1 == 0:
                break
        assert r0 == 0 and r1 == 0
        for k in range(2):
            ocam = oracle.camera(eyes[k], look)
            want = o_s.render(4, ocam, (oracle.Light * 2)(oracle.light(lp, ocam)), 1, oracle.default_opts(W, H))[0]
            assert np.array_equal(bufs[k].cpu().numpy().view(np.uint32), want), k
    finally:
        L.mi355_mgpu_destroy(m)


def test_config5_frame_on_eight_virtual_ranks(oracle):
    """BASELINE config 5's frame -- dragon, 3840x2160, depth 3 -- cut over eight ranks (one GPU plays them all), assembled on
    rank 0, against the oracle's frame."""
    torch = pytest.importorskip("torch")
    L = _mgpu_api()
    W, H = 3840, 2160
    s = R.Scene(R.assets.mesh_path("dragon_vis.ply"))
    s.bvh_create()
    nodes, idx = s.bvh_arrays()
    m = L.mi355_mgpu_create(C.byref(s.desc), (C.c_int * 8)(*([0] * 8)), 8)
    assert m, L.mi355_last_error()
    try:
        assert L.mi355_mgpu_set_bvh(m, nodes.ctypes.data, nodes.shape[0], idx.ctypes.data, idx.shape[0]) == 0
        cam, lights, n = R.benchmark_frame(0)
        o = R.default_opts(W, H)
        buf = torch.zeros((H, W), dtype=torch.int32, device="cuda:0")
        outs = (C.c_void_p * 1)(buf.data_ptr())
        t, st = C.c_int(0), R.Stats()
        assert L.mi355_mgpu_render_batch(m, 9, 1, C.byref(cam), lights, n, C.byref(o), outs, W * 4, C.byref(t)) == 0, L.mi355_last_error()
        assert L.mi355_mgpu_wait(m, t.value, C.byref(st)) == 0, L.mi355_last_error()
    finally:
        L.mi355_mgpu_destroy(m)
    osc = oracle.Scene(R.assets.mesh_path("dragon_vis.ply"))
    osc.bvh_build()
    ocam, olights, on = oracle.benchmark_frame(0)
    import os
    ref, _, ost = osc.render(9, ocam, olights, on, oracle.default_opts(W, H, threads=os.cpu_count() or 1))
    assert np.array_equal(buf.cpu().numpy().view(np.uint32), ref)
    assert (st.normal_rays, st.shadow_rays) == (ost.normal_rays, ost.shadow_rays) == (9410297, 1163742)      # SURVEY 8(d), cfg 5


def test_render_cli_on_two_ranks_and_bench_statistics(tmp_path):
    import os
    cli = os.path.join(os.path.dirname(R.RENDER_SO), "render_cli")
    mesh = R.assets.mesh_path("trainColor.tri")
    out = subprocess.run([cli, "-b", "-n", "8", "-m", "8", "-g", "0,0", "-o", str(tmp_path / "two"), mesh], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    ref = subprocess.run([cli, "-b", "-n", "8", "-m", "8", "-o", str(tmp_path / "one"), mesh], capture_output=True, text=True, timeout=300)
    assert ref.returncode == 0, ref.stderr
    for f in (1, 8):
        assert open(str(tmp_path / ("two_%04d.ppm" % f)), "rb").read() == open(str(tmp_path / ("one_%04d.ppm" % f)), "rb").read()
    b = subprocess.run([cli, "--bench", "-n", "20", mesh], capture_output=True, text=True, timeout=300)     # `make bench` with shorter runs
    assert b.returncode == 0, b.stderr
    lines = b.stdout.strip().splitlines()
    assert [l.split(":")[0].strip() for l in lines[-5:]] == ["Average value", "Std deviation", "Median", "Min", "Max"]
    assert b.stdout.count("Rendering 20 frames in") == 5
