"""MLAA on the device (mi355_opts.mlaa, mi355_mlaa_device): the reference's post filter (MLAA.cc, pinned to the real
code on the CPU: tests/test_refcore_pins.py) against the oracle, on frames of every mode and on noise."""
import ctypes as C

import numpy as np
import pytest

import renderer_amd as R

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def mlaa_device(s, img):
    dev = torch.device("cuda", 0)
    t = torch.from_numpy(img.astype(np.int32)).to(dev)
    f = R.lib().mi355_mlaa_device
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    rc = f(s.context(), t.data_ptr(), img.shape[1] * 4, img.shape[0], torch.cuda.current_stream(dev).cuda_stream)
    assert rc == 0, R.lib().mi355_last_error()
    torch.cuda.synchronize(dev)
    return t.cpu().numpy().astype(np.uint32)


def test_mlaa_on_noise_and_patterns(oracle):
    s = R.Scene(R.assets.mesh_path("chessboard.tri"))
    rng = np.random.default_rng(8)
    for it in range(40):
        W, H = int(rng.integers(2, 60)) * 4, int(rng.integers(1, 30)) * 8
        kind = it % 4
        if kind == 0:
            img = rng.integers(0, 1 << 24, (H, W), dtype=np.uint32)
        elif kind == 1:
            img = (rng.integers(0, 3, (H, W)) * 0x404040).astype(np.uint32)
        elif kind == 2:
            img = np.zeros((H, W), np.uint32)
            for _ in range(6):
                x0, y0 = int(rng.integers(0, W)), int(rng.integers(0, H))
                img[y0:y0 + int(rng.integers(1, H)), x0:x0 + int(rng.integers(1, W))] = int(rng.integers(0, 1 << 24))
        else:
            yy, xx = np.mgrid[0:H, 0:W]
            img = (((xx * int(rng.integers(1, 5)) + yy * int(rng.integers(1, 5))) // int(rng.integers(3, 17))) % 2 * 0xffffff).astype(np.uint32)
        assert np.array_equal(mlaa_device(s, img), oracle.mlaa(img)), "case %d (%dx%d)" % (it, W, H)
    f = R.lib().mi355_mlaa_device
    assert f(s.context(), C.c_void_p(16), 10 * 4, 16, None) == -20        # width not a multiple of four
    # a surface wider or taller than the 16384 pixels the scan's LDS list of line starts is sized for: refused, not written
    assert f(s.context(), C.c_void_p(16), 16388 * 4, 16, None) == -20
    assert f(s.context(), C.c_void_p(16), 16 * 4, 16392, None) == -20


@pytest.mark.parametrize("mode,mesh,W,H", [(6, "chessboard.tri", 1920, 1080), (8, "chessboard.tri", 800, 600), (9, "dragon_vis.ply", 800, 600),
                                          (2, "chessboard.tri", 640, 480), (4, "dragon_vis.ply", 640, 360)])
def test_rendered_frames_with_the_mlaa_option(oracle, oracle_scene, mode, mesh, W, H):
    hs, osc = R.Scene(R.assets.mesh_path(mesh)), oracle_scene(mesh, mode >= 9)
    if mode >= 9:
        hs.bvh_create()
    cam, lights, n = R.benchmark_frame(7)
    ocam, olights, on = oracle.benchmark_frame(7)
    maps = None
    if mode in (7, 8):
        maps = [osc.shadowmap(olights[0])]
        hs.shadowmap_render(0, lights[0])
    plain = osc.render(mode, ocam, olights, on, oracle.default_opts(W, H, threads=8), shadow_maps=maps)[0]
    want = oracle.mlaa(plain)
    got = hs.render(mode, cam, lights, n, R.default_opts(W, H, mlaa=1))[0]
    assert (want != plain).sum() > 50
    assert np.array_equal(got, want)
    # pipelined and batched frames take the same filter
    buf = np.zeros((H, W), np.uint32)
    hs.render_wait(hs.render_async(mode, cam, lights, n, R.default_opts(W, H, mlaa=1), buf))
    assert np.array_equal(buf, want)
    if mode in (6, 8, 9):
        dev = torch.device("cuda", 0)
        outs = [torch.zeros((H, W), dtype=torch.int32, device=dev) for _ in range(2)]
        hs.render_batch_device(mode, [cam, cam], [lights, lights], n, R.default_opts(W, H, mlaa=1), [o.data_ptr() for o in outs], W * 4, None,
                               torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize(dev)
        for o in outs:
            assert np.array_equal(o.cpu().numpy().astype(np.uint32), want)
    with pytest.raises(R.Mi355Error, match="mlaa"):
        hs.render(mode, cam, lights, n, R.default_opts(W, H - 3, mlaa=1))
