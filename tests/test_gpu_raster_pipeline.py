"""Raster frames of the device entry points are pipelined inside the library: setup and fill of frame n+1 run on an internal
stream beside the tile kernel of frame n (two scratch sets, events).  Whatever the caller does between frames -- same output
buffer, other buffers, other sizes and modes, counting frames, batches, a second stream -- every frame must be the frame the
one-stream path (tune flag 32) draws."""
import numpy as np
import pytest

import renderer_amd as R

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def scene():
    s = R.Scene(R.assets.mesh_path("chessboard.tri"))
    cam, lights, n = R.benchmark_frame(0)
    s.shadowmap_render(0, lights[0])
    return s


def reference(scene, mode, k, W, H):
    cam, lights, n = R.benchmark_frame(k)
    return scene.render(mode, cam, lights, n, R.default_opts(W, H, tune=R.tune(nopipe=1)))[0]


def test_back_to_back_frames_without_a_sync_in_between(scene):
    W, H, N = 800, 600, 24
    stream = torch.cuda.current_stream()
    bufs = torch.full((N, H, W), 0x777777, dtype=torch.int32, device="cuda")
    one = torch.zeros((H, W), dtype=torch.int32, device="cuda")
    keep = []
    for k in range(N):
        cam, lights, n = R.benchmark_frame(3 * k)
        mode = (6, 8, 4, 7, 5)[k % 5]
        scene.render_device(mode, cam, lights, n, R.default_opts(W, H), bufs[k].data_ptr(), W * 4, 0, stream.cuda_stream)
        # ... and the same frame into ONE buffer that the caller copies away in stream order before the next frame lands in it
        scene.render_device(mode, cam, lights, n, R.default_opts(W, H), one.data_ptr(), W * 4, 0, stream.cuda_stream)
        keep.append(one.clone())
    torch.cuda.synchronize()
    got = bufs.cpu().numpy().view(np.uint32)
    for k in range(N):
        ref = reference(scene, (6, 8, 4, 7, 5)[k % 5], 3 * k, W, H)
        assert np.array_equal(got[k], ref), "frame %d" % k
        assert np.array_equal(keep[k].cpu().numpy().view(np.uint32), ref), "frame %d through the shared buffer" % k


def test_sizes_counting_frames_and_batches_between_pipelined_frames(scene):
    stream = torch.cuda.current_stream()
    cases = [(640, 360, 6), (1920, 1080, 8), (333, 187, 6), (640, 360, 4), (1280, 720, 8)]
    outs = []
    for i in range(15):
        W, H, mode = cases[i % len(cases)]
        cam, lights, n = R.benchmark_frame(7 * i)
        buf = torch.zeros((H, W), dtype=torch.int32, device="cuda")
        scene.render_device(mode, cam, lights, n, R.default_opts(W, H), buf.data_ptr(), W * 4, 0, stream.cuda_stream)
        outs.append((buf, mode, 7 * i, W, H))
        if i % 4 == 1:      # a counting frame (outside the pipeline, same scratch) right behind it
            scene.render_device(mode, cam, lights, n, R.default_opts(W, H, collect_stats=1), buf.data_ptr(), W * 4, 0, stream.cuda_stream)
        if i % 4 == 2:      # a batch
            b2 = torch.zeros((2, H, W), dtype=torch.int32, device="cuda")
            cs = [R.benchmark_frame(7 * i + j) for j in (1, 2)]
            scene.render_batch_device(mode, [c[0] for c in cs], [c[1] for c in cs], 1, R.default_opts(W, H), [b2[j].data_ptr() for j in range(2)], W * 4, None, stream.cuda_stream)
            outs.append((b2[0], mode, 7 * i + 1, W, H)); outs.append((b2[1], mode, 7 * i + 2, W, H))
    torch.cuda.synchronize()
    for buf, mode, k, W, H in outs:
        assert np.array_equal(buf.cpu().numpy().view(np.uint32), reference(scene, mode, k, W, H)), "frame %d mode %d %dx%d" % (k, mode, W, H)


def test_frames_on_two_streams(scene):
    """the tile kernels follow the caller's stream, whichever it is"""
    W, H = 640, 360
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    bufs = torch.zeros((12, H, W), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for k in range(12):
        cam, lights, n = R.benchmark_frame(5 * k)
        st = s1 if k % 2 == 0 else s2
        scene.render_device(6, cam, lights, n, R.default_opts(W, H), bufs[k].data_ptr(), W * 4, 0, st.cuda_stream)
        st.synchronize()          # (one context, one default control block: frames of different streams do not overlap)
    got = bufs.cpu().numpy().view(np.uint32)
    for k in range(12):
        assert np.array_equal(got[k], reference(scene, 6, 5 * k, W, H))


def test_padded_pitch_and_bands(scene):
    """overlapped frames go through a buffer of the library's with the caller's pitch: the caller's padding words stay as they
    were, banded frames (interleaved rows, compact or in place) land where the one-stream path puts them"""
    W, H, PAD = 500, 300, 12
    stream = torch.cuda.current_stream()
    tn = R.tune()
    jobs = []
    for i in range(10):
        cam, lights, n = R.benchmark_frame(11 * i)
        mode = (6, 8)[i % 2]
        kw = {}
        if i % 3 == 1: kw = dict(band_rows=8, band_index=i % 2, band_count=2, compact_rows=1)
        if i % 3 == 2: kw = dict(band_rows=8, band_index=1, band_count=3, compact_rows=0)
        rows = H
        if kw.get("compact_rows"):
            rows = sum(1 for y in range(H) if (y // 8) % 2 == kw["band_index"])
        a = torch.full((rows, W + PAD), 0x5a5a5a, dtype=torch.int32, device="cuda")
        b = torch.full((rows, W + PAD), 0x5a5a5a, dtype=torch.int32, device="cuda")
        scene.render_device(mode, cam, lights, n, R.default_opts(W, H, tune=tn, **kw), a.data_ptr(), (W + PAD) * 4, 0, stream.cuda_stream)
        jobs.append((a, b, mode, cam, lights, n, kw))
    torch.cuda.synchronize()
    for a, b, mode, cam, lights, n, kw in jobs:
        scene.render_device(mode, cam, lights, n, R.default_opts(W, H, tune=R.tune(nopipe=1), **kw), b.data_ptr(), (W + PAD) * 4, 0, stream.cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(a, b), "mode %d %r" % (mode, kw)
        assert bool((a[:, W:] == 0x5a5a5a).all())
