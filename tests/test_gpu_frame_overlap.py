"""Raytraced frames of mi355_render_device overlap inside the library like the raster frames (test_gpu_raster_pipeline.py): up
to three run on internal streams, each with its own control block and culled-tile list, into buffers of the library's; the
caller's stream copies them out.  The caller sees frames in stream order, whatever else it enqueues in between, and
mi355_fetch_stats describes the last frame."""
import numpy as np
import pytest

import renderer_amd as R

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def scene():
    s = R.Scene(R.assets.mesh_path("statue.ply"))
    s.bvh_create()
    cam, lights, n = R.benchmark_frame(0)
    s.shadowmap_render(0, lights[0])
    return s


def host_frame(scene, mode, k, W, H, **kw):
    cam, lights, n = R.benchmark_frame(k)
    px, _, st = scene.render(mode, cam, lights, n, R.default_opts(W, H, **kw))
    return px, st


def test_back_to_back_raytraced_frames_and_their_counters(scene):
    W, H, N = 480, 270, 14
    stream = torch.cuda.current_stream()
    one = torch.full((H, W), 0x123456, dtype=torch.int32, device="cuda")
    keep = []
    for k in range(N):
        cam, lights, n = R.benchmark_frame(9 * k)
        scene.render_device(9, cam, lights, n, R.default_opts(W, H), one.data_ptr(), W * 4, 0, stream.cuda_stream)
        keep.append(one.clone())           # (in stream order: before the next frame lands in the buffer)
    torch.cuda.synchronize()
    last = scene.fetch_stats()
    for k in range(N):
        ref, st = host_frame(scene, 9, 9 * k, W, H)
        assert np.array_equal(keep[k].cpu().numpy().view(np.uint32), ref), "frame %d" % k
    assert (last.normal_rays, last.shadow_rays) == (st.normal_rays, st.shadow_rays)


def test_modes_sizes_filters_and_other_calls_in_between(scene):
    stream = torch.cuda.current_stream()
    cases = [(9, 320, 200, {}), (6, 640, 360, {}), (10, 200, 120, {}), (9, 333, 187, {}), (9, 320, 200, dict(mlaa=1)),
             (8, 320, 200, {}), (9, 320, 200, dict(use_shadows=0)), (9, 256, 256, dict(max_ray_depth=1))]
    outs = []
    for i in range(20):
        mode, W, H, kw = cases[i % len(cases)]
        cam, lights, n = R.benchmark_frame(13 * i)
        buf = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        scene.render_device(mode, cam, lights, n, R.default_opts(W, H, **kw), buf.data_ptr(), W * 4, 0, stream.cuda_stream)
        outs.append((buf, mode, 13 * i, W, H, kw))
        if i % 5 == 1:       # a counting frame on the same stream (not overlapped: it uses the context's control block)
            scene.render_device(mode, cam, lights, n, R.default_opts(W, H, collect_stats=1, **kw), buf.data_ptr(), W * 4, 0, stream.cuda_stream)
        if i % 5 == 3 and mode == 9:       # a batch of two
            b2 = torch.zeros((2, H, W), dtype=torch.int32, device="cuda")
            cs = [R.benchmark_frame(13 * i + j) for j in (1, 2)]
            scene.render_batch_device(mode, [c[0] for c in cs], [c[1] for c in cs], 1, R.default_opts(W, H, **kw), [b2[j].data_ptr() for j in range(2)], W * 4, None, stream.cuda_stream)
            outs.append((b2[0], mode, 13 * i + 1, W, H, kw)); outs.append((b2[1], mode, 13 * i + 2, W, H, kw))
    torch.cuda.synchronize()
    for buf, mode, k, W, H, kw in outs:
        assert np.array_equal(buf.cpu().numpy().view(np.uint32), host_frame(scene, mode, k, W, H, **kw)[0]), "frame %d mode %d %dx%d %r" % (k, mode, W, H, kw)


def test_padding_bands_and_float_output(scene):
    """padded pitch: the caller's padding words stay; compact bands overlap, bands in place and frames with a float buffer take
    the one-stream path -- all of them equal that path's frames (tune flag 32)"""
    W, H, PAD = 400, 240, 8
    stream = torch.cuda.current_stream()
    jobs = []
    for i in range(9):
        cam, lights, n = R.benchmark_frame(17 * i)
        kw = {}
        if i % 3 == 1: kw = dict(band_rows=8, band_index=i % 2, band_count=2, compact_rows=1)
        if i % 3 == 2: kw = dict(band_rows=8, band_index=1, band_count=3, compact_rows=0)
        rows = sum(1 for y in range(H) if (y // 8) % 2 == kw["band_index"]) if kw.get("compact_rows") else H
        a = torch.full((rows, W + PAD), 0x5a5a5a, dtype=torch.int32, device="cuda")
        b = torch.full((rows, W + PAD), 0x5a5a5a, dtype=torch.int32, device="cuda")
        af = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") if i == 6 else None
        scene.render_device(9, cam, lights, n, R.default_opts(W, H, **kw), a.data_ptr(), (W + PAD) * 4, af.data_ptr() if af is not None else 0, stream.cuda_stream)
        jobs.append((a, b, cam, lights, n, kw))
    torch.cuda.synchronize()
    for a, b, cam, lights, n, kw in jobs:
        scene.render_device(9, cam, lights, n, R.default_opts(W, H, tune=R.tune(nopipe=1), **kw), b.data_ptr(), (W + PAD) * 4, 0, stream.cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(a, b), repr(kw)
        assert bool((a[:, W:] == 0x5a5a5a).all())


def test_frames_of_two_streams_and_a_new_tree_in_between(scene):
    W, H = 320, 200
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    bufs = torch.zeros((10, H, W), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for k in range(10):
        cam, lights, n = R.benchmark_frame(21 * k)
        st = s1 if k % 2 == 0 else s2
        scene.render_device(9, cam, lights, n, R.default_opts(W, H), bufs[k].data_ptr(), W * 4, 0, st.cuda_stream)
        if k == 5:
            torch.cuda.synchronize()
            scene.build_bvh_device()         # the same tree again, built on the device
    torch.cuda.synchronize()
    got = bufs.cpu().numpy().view(np.uint32)
    for k in range(10):
        assert np.array_equal(got[k], host_frame(scene, 9, 21 * k, W, H)[0]), "frame %d" % k


def test_consecutive_batches_into_the_same_buffers(scene):
    """batches overlap too: the caller's buffers hold batch i after call i in stream order, although batch i + 1 is already running"""
    W, H, NB = 384, 216, 3
    stream = torch.cuda.current_stream()
    bufs = torch.zeros((NB, H, W), dtype=torch.int32, device="cuda")
    keep, rays = [], []
    for i in range(7):
        ks = [5 * (NB * i + j) for j in range(NB)]
        cs = [R.benchmark_frame(k) for k in ks]
        mode = 10 if i == 3 else 9
        scene.render_batch_device(mode, [c[0] for c in cs], [c[1] for c in cs], 1, R.default_opts(W, H), [bufs[j].data_ptr() for j in range(NB)], W * 4, None, stream.cuda_stream)
        keep.append((bufs.clone(), mode, ks))
        if i == 4:           # a single frame between two batches
            cam, lights, n = R.benchmark_frame(3)
            scene.render_device(9, cam, lights, n, R.default_opts(W, H), bufs[1].data_ptr(), W * 4, 0, stream.cuda_stream)
            keep.append((bufs[1:2].clone(), 9, [3]))
    torch.cuda.synchronize()
    last = scene.fetch_stats()
    total = 0
    for got, mode, ks in keep:
        total = 0
        for j, k in enumerate(ks):
            ref, st = host_frame(scene, mode, k, W, H)
            total += st.normal_rays + st.shadow_rays
            assert np.array_equal(got[j].cpu().numpy().view(np.uint32), ref), "mode %d frame %d" % (mode, k)
    assert last.normal_rays + last.shadow_rays == total        # the counters are those of the last call

