"""Context lifecycle on the device: scenes created, used and destroyed in a loop must give all their device memory back
(mi355_scene_destroy frees every stream, including the batch scratch and the builder's buffers), and several contexts
can be alive and rendering at once."""
import gc

import numpy as np
import pytest

import renderer_amd as R

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def one_round(mesh, modes):
    s = R.Scene(R.assets.mesh_path(mesh))
    s.bvh_create("device")
    cam, lights, n = R.benchmark_frame(5)
    out = []
    for mode in modes:
        if mode in (7, 8):
            s.shadowmap_render(0, lights[0])
        out.append(s.render(mode, cam, lights, n, R.default_opts(320, 240))[0])
    dev = torch.device("cuda", 0)
    bufs = [torch.zeros((240, 320), dtype=torch.int32, device=dev) for _ in range(3)]
    cl = [R.benchmark_frame(f) for f in (1, 2, 3)]
    for mode in (9, 6):
        s.render_batch_device(mode, [c[0] for c in cl], [c[1] for c in cl], 1, R.default_opts(320, 240),
                              [b.data_ptr() for b in bufs], 320 * 4, None, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    return out


def free_bytes():
    gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
    return torch.cuda.mem_get_info(0)[0]


def test_destroy_returns_all_device_memory():
    first = [a.copy() for a in one_round("dragon_vis.ply", (9, 6, 8, 2))]      # (copies: views keep their Scene alive)
    one_round("chessboard.tri", (9, 6, 8, 2))             # warm-up: the runtime's pools reach their size once
    free0 = free_bytes()
    for i in range(24):
        again = one_round("dragon_vis.ply" if i % 2 else "chessboard.tri", (9, 6, 8, 2))
        if i % 2:
            assert all(np.array_equal(a, b) for a, b in zip(first, again))
        del again
    lost = free0 - free_bytes()
    assert lost < 4 << 20, "device memory shrank by %.1f MB over 24 create/destroy rounds" % (lost / 2**20)


def test_several_contexts_alive_at_once():
    scenes = [R.Scene(R.assets.mesh_path(m)) for m in ("dragon_vis.ply", "chessboard.tri", "dragon_vis.ply")]
    for s in scenes:
        s.bvh_create()
    cam, lights, n = R.benchmark_frame(9)
    a = [s.render(9, cam, lights, n, R.default_opts(200, 150))[0] for s in scenes]
    b = [s.render(9, cam, lights, n, R.default_opts(200, 150))[0] for s in reversed(scenes)][::-1]
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert np.array_equal(a[0], a[2]) and not np.array_equal(a[0], a[1])
