"""Bad arguments at the C ABI must come back as error codes with a message (int < 0 + mi355_last_error), never as a crash
or a silently wrong frame: the reference's callers cannot fail, so the replacement must say when it does."""
import ctypes as C

import numpy as np
import pytest

import renderer_amd as R

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def scene():
    s = R.Scene(R.assets.mesh_path("chessboard.tri"))
    s.bvh_create()
    return s


@pytest.mark.parametrize("kw", [dict(width=0), dict(height=-5), dict(width=20000), dict(screen_dist=0), dict(max_ray_depth=0),
                                dict(max_ray_depth=9), dict(band_count=3, band_index=3, band_rows=8), dict(band_count=2, band_rows=0),
                                dict(shadowmap_size=0)], ids=lambda k: ",".join("%s=%s" % kv for kv in k.items()))
def test_bad_options_are_refused(scene, kw):
    cam, lights, n = R.benchmark_frame(0)
    o = R.default_opts(64, 48)
    for k, v in kw.items():
        setattr(o, k, v)
    out = np.zeros((256, 256), np.uint32)                  # straight at the C ABI: the binding would trip over the sizes first
    L = R.lib()
    assert L.mi355_render(scene.context(), 9, C.byref(cam), lights, n, C.byref(o), out.ctypes.data, 256 * 4, None, None) < 0
    assert len(L.mi355_last_error()) > 0
    assert not out.any()


@pytest.mark.parametrize("mode", [0, 11, -1, 100])
def test_unknown_or_unsupported_modes_are_refused(scene, mode):
    cam, lights, n = R.benchmark_frame(0)
    with pytest.raises(R.Mi355Error):
        scene.render(mode, cam, lights, n, R.default_opts(64, 48))


def test_null_pointers_and_bad_counts(scene):
    L = R.lib()                                            # argtypes as the binding declares them; None = NULL
    ctx = scene.context()
    cam, lights, n = R.benchmark_frame(0)
    o = R.default_opts(64, 48)
    out = np.zeros((48, 64), np.uint32)
    good = dict(ctx=ctx, cam=C.byref(cam), lights=lights, n=1, o=C.byref(o), out=out.ctypes.data)

    def call(**over):
        a = dict(good, **over)
        return L.mi355_render(a["ctx"], 9, a["cam"], a["lights"], a["n"], a["o"], a["out"], 64 * 4, None, None)
    assert call() == 0
    for over in (dict(ctx=None), dict(cam=None), dict(o=None), dict(out=None), dict(lights=None), dict(n=5), dict(n=-1)):
        assert call(**over) < 0, over
        assert len(L.mi355_last_error()) > 0
    assert call() == 0                                     # the context is still usable after the refusals


def test_batch_limits(scene):
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev).cuda_stream
    W, H = 64, 48
    cl = [R.benchmark_frame(f) for f in range(70)]
    bufs = [torch.zeros((H, W), dtype=torch.int32, device=dev) for _ in range(70)]
    for n in (0, 65, 70):
        with pytest.raises(R.Mi355Error):
            scene.render_batch_device(9, [c[0] for c in cl[:n]], [c[1] for c in cl[:n]], 1, R.default_opts(W, H),
                                      [b.data_ptr() for b in bufs[:n]], W * 4, None, st)
    scene.render_batch_device(9, [c[0] for c in cl[:64]], [c[1] for c in cl[:64]], 1, R.default_opts(W, H),
                              [b.data_ptr() for b in bufs[:64]], W * 4, None, st)
    torch.cuda.synchronize(dev)
    one = scene.render(9, cl[63][0], cl[63][1], 1, R.default_opts(W, H))[0]
    assert np.array_equal(bufs[63].cpu().numpy().astype(np.uint32), one)


def test_raytrace_without_a_bvh_is_refused():
    s = R.Scene(R.assets.mesh_path("chessboard.tri"))
    cam, lights, n = R.benchmark_frame(0)
    with pytest.raises(R.Mi355Error):
        s.render(9, cam, lights, n, R.default_opts(64, 48))
    assert s.render(6, cam, lights, n, R.default_opts(64, 48))[0].any()       # raster modes need none
