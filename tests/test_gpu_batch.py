"""mi355_render_batch_device: several frames of the orbit in one launch.  Every frame must be exactly the frame a
single-frame launch produces (and therefore the oracle's): the batch only changes the schedule."""
import os

import numpy as np
import pytest

import renderer_amd as R

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dragon():
    s = R.Scene(R.assets.mesh_path("dragon_vis.ply"))
    s.bvh_update()
    return s


def render_single(s, mode, frames, opts, W, H, want_f32=False):
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev)
    outs, outfs, rays = [], [], 0
    for f in frames:
        cam, lights, n = R.benchmark_frame(f)
        b = torch.zeros((H, W), dtype=torch.int32, device=dev)
        bf = torch.zeros((H, W, 3), dtype=torch.float32, device=dev) if want_f32 else None
        s.render_device(mode, cam, lights, n, opts, b.data_ptr(), W * 4, bf.data_ptr() if want_f32 else 0, st.cuda_stream)
        torch.cuda.synchronize(dev)
        stt = s.fetch_stats()
        rays += stt.normal_rays + stt.shadow_rays
        outs.append(b); outfs.append(bf)
    return outs, outfs, rays


def render_batch(s, mode, frames, opts, W, H, want_f32=False, second_light=False):
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev)
    cl = [R.benchmark_frame(f, second_light) for f in frames]
    bufs = [torch.zeros((H, W), dtype=torch.int32, device=dev) for _ in frames]
    bfs = [torch.zeros((H, W, 3), dtype=torch.float32, device=dev) for _ in frames] if want_f32 else None
    s.render_batch_device(mode, [c[0] for c in cl], [c[1] for c in cl], cl[0][2], opts, [b.data_ptr() for b in bufs], W * 4,
                          [b.data_ptr() for b in bfs] if want_f32 else None, st.cuda_stream)
    torch.cuda.synchronize(dev)
    stt = s.fetch_stats()
    return bufs, bfs, stt.normal_rays + stt.shadow_rays


@pytest.mark.parametrize("n", [2, 3, 8, 20])
def test_batch_equals_single_frames(dragon, n):
    W, H = 640, 360
    o = R.default_opts(W, H)
    frames = [5 + 7 * j for j in range(n)]
    a, af, rays_a = render_single(dragon, 9, frames, o, W, H, want_f32=True)
    b, bf, rays_b = render_batch(dragon, 9, frames, o, W, H, want_f32=True)
    for j in range(n):
        assert bool((a[j] == b[j]).all()), "frame %d of the batch differs" % j
        assert bool((af[j] == bf[j]).all())
        assert int((b[j] != 0).sum()) > 1000
    assert rays_a == rays_b                       # ray counters of a batch are totals


def test_batch_full_size_matches_the_reference_frame_hash(dragon):
    """Frame f0 inside a 1080p batch of four hashes to the REAL reference's pin."""
    import hashlib, json
    pins = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_pins.json")))
    pin = [p for p in pins["frames"] if p["id"] == "cfg4"][0]
    W, H = pin["w"], pin["h"]
    b, _, _ = render_batch(dragon, 9, [3, 0, 150, 77], R.default_opts(W, H), W, H)
    img = b[1].cpu().numpy().astype(np.uint32)
    assert hashlib.sha256(R.rgb_bytes(img)).hexdigest() == pin["sha256"]


def test_batch_antialias_two_lights_and_knobs(oracle, oracle_scene):
    s = R.Scene(R.assets.mesh_path("chessboard.tri"))
    s.bvh_update()
    W, H = 320, 240
    for mode, second, tune in ((10, False, {}), (9, True, {}), (9, False, dict(xmin=1, rmin=1)), (9, False, dict(exact=1)),
                               (9, False, dict(nohelp=1, bpc=1)), (9, True, dict(bpc=3)), (10, False, dict(bpc=4)), (9, True, dict(bpc=4, exact=1))):
        o = R.default_opts(W, H, tune=tune)
        frames = [2, 40, 41]
        dev = torch.device("cuda", 0)
        b, _, _ = render_batch(s, mode, frames, o, W, H, second_light=second)
        osc = oracle_scene("chessboard.tri", True)
        for j, f in enumerate(frames):
            ocam, ol, on = oracle.benchmark_frame(f, second)
            want = osc.render(mode, ocam, ol, on, oracle.default_opts(W, H, threads=os.cpu_count() or 1))[0]
            assert np.array_equal(b[j].cpu().numpy().astype(np.uint32), want), "mode %d frame %d tune %s" % (mode, f, tune)


def test_batch_with_band_sharding(dragon):
    """The multi-GPU path: each rank renders its bands of every frame of the batch into compact buffers."""
    W, H = 480, 270
    frames = [0, 9, 18]
    full, _, _ = render_batch(dragon, 9, frames, R.default_opts(W, H), W, H)
    from renderer_amd import multigpu
    parts = []
    for rank in range(3):
        o = R.default_opts(W, H, band_rows=multigpu.BAND_ROWS, band_index=rank, band_count=3, compact_rows=1)
        b, _, _ = render_batch(dragon, 9, frames, o, W, H)
        rows = multigpu.rows_of_rank(H, multigpu.BAND_ROWS, 3, rank)
        parts.append([x.cpu().numpy()[:rows] for x in b])
    for j in range(len(frames)):
        got = multigpu.assemble_numpy([parts[r][j] for r in range(3)], H, multigpu.BAND_ROWS)
        assert np.array_equal(got, full[j].cpu().numpy())


@pytest.mark.parametrize("mode", [6, 8])
def test_raster_batch_with_band_sharding(mode):
    """What bench.py --gpus N times for the rasterizer (multi_gpu.raster_1080p): every rank rasterizes its interleaved bands of all
    frames of a step in one batched launch into compact buffers; assembled, they are the frames a single device draws."""
    from renderer_amd import multigpu
    s = R.Scene(R.assets.mesh_path("chessboard.tri"))
    s.shadowmap_render(0, R.benchmark_frame(0)[1][0])
    W, H = 642, 363                                    # 46 bands: the ranks own different numbers of rows, the last band is short
    frames = [0, 37, 100, 150, 199]
    full, _, _ = render_batch(s, mode, frames, R.default_opts(W, H), W, H)
    for world in (2, 3):
        parts = []
        for rank in range(world):
            o = R.default_opts(W, H, band_rows=multigpu.BAND_ROWS, band_index=rank, band_count=world, compact_rows=1)
            b, _, _ = render_batch(s, mode, frames, o, W, H)
            rows = multigpu.rows_of_rank(H, multigpu.BAND_ROWS, world, rank)
            parts.append([x.cpu().numpy()[:rows] for x in b])
        for j in range(len(frames)):
            got = multigpu.assemble_numpy([parts[r][j] for r in range(world)], H, multigpu.BAND_ROWS)
            assert np.array_equal(got, full[j].cpu().numpy()), "mode %d, %d ranks, frame %d" % (mode, world, frames[j])


def test_batch_argument_errors(dragon):
    W, H = 64, 48
    dev = torch.device("cuda", 0)
    buf = torch.zeros((H, W), dtype=torch.int32, device=dev)
    cam, lights, n = R.benchmark_frame(0)
    with pytest.raises(R.Mi355Error, match="raster and raytrace modes only"):
        dragon.render_batch_device(2, [cam, cam], [lights, lights], n, R.default_opts(W, H), [buf.data_ptr()] * 2, W * 4)
    with pytest.raises(R.Mi355Error, match="n_frames"):
        dragon.render_batch_device(9, [cam] * 65, [lights] * 65, n, R.default_opts(W, H), [buf.data_ptr()] * 65, W * 4)
    with pytest.raises(R.Mi355Error, match="cannot collect"):
        dragon.render_batch_device(9, [cam, cam], [lights, lights], n, R.default_opts(W, H, collect_stats=1), [buf.data_ptr()] * 2, W * 4)
    with pytest.raises(R.Mi355Error, match="ordered walk"):
        dragon.render_batch_device(9, [cam, cam], [lights, lights], n, R.default_opts(W, H, tune=dict(reforder=1)), [buf.data_ptr()] * 2, W * 4)
    torch.cuda.synchronize(dev)


@pytest.mark.parametrize("mode", [4, 5, 6, 7, 8])
def test_raster_batch_equals_single_frames(mode):
    """Raster modes: the frames of a batch run side by side on internal streams, each with its own scratch; every frame
    is bit for bit the single-frame render (which the parity tests compare with the oracle)."""
    s = R.Scene(R.assets.mesh_path("chessboard.tri"))
    W, H = 800, 600
    dev = torch.device("cuda", 0)
    st = torch.cuda.current_stream(dev)
    o = R.default_opts(W, H)
    frames = [1, 30, 31, 77, 150]
    if mode in (7, 8):
        s.shadowmap_render(0, R.benchmark_frame(0)[1][0])
    single = []
    for f in frames:
        cam, lights, n = R.benchmark_frame(f)
        b = torch.zeros((H, W), dtype=torch.int32, device=dev)
        s.render_device(mode, cam, lights, n, o, b.data_ptr(), W * 4, 0, st.cuda_stream)
        single.append(b)
    for rep in range(3):                  # repeated: the frames share nothing but the scene and the shadow map
        bufs = [torch.full((H, W), 12345, dtype=torch.int32, device=dev) for _ in frames]
        cl = [R.benchmark_frame(f) for f in frames]
        s.render_batch_device(mode, [c[0] for c in cl], [c[1] for c in cl], cl[0][2], o, [b.data_ptr() for b in bufs], W * 4, None, st.cuda_stream)
        torch.cuda.synchronize(dev)
        for j in range(len(frames)):
            assert bool((single[j] == bufs[j]).all()), "mode %d: frame %d of the batch differs (repeat %d)" % (mode, j, rep)
            assert int((bufs[j] != 0).sum()) > 1000
