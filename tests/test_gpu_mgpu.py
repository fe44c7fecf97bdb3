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
