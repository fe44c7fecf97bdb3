"""Thread compatibility of the C ABI (one host thread per context): several host threads, each with its own
context, render different modes at the same time; every frame must be the frame a lone thread gets."""
import threading

import numpy as np
import pytest

import renderer_amd as R

pytestmark = pytest.mark.gpu


def test_four_threads_four_contexts():
    meshes = ["dragon_vis.ply", "chessboard.tri", "legocar.3ds", "dragon_vis.ply"]
    modes = [(9, 6, 2), (6, 8, 4), (9, 10, 5), (7, 9, 1)]
    sizes = [(320, 240), (333, 217), (200, 150), (641, 97)]
    scenes = []
    for m in meshes:
        s = R.Scene(R.assets.mesh_path(m))
        s.bvh_create()
        scenes.append(s)
    cams = [R.benchmark_frame(k) for k in range(12)]

    def frames(i, rounds):
        s, out = scenes[i], []
        W, H = sizes[i]
        for r in range(rounds):
            for mode in modes[i]:
                cam, lights, n = cams[(3 * r + i) % 12]
                if mode in (7, 8):
                    s.shadowmap_render(0, lights[0])
                out.append(s.render(mode, cam, lights, n, R.default_opts(W, H))[0].copy())
        return out

    want = [frames(i, 6) for i in range(4)]                # one after the other
    got, errs = [None] * 4, []

    def work(i):
        try:
            got[i] = frames(i, 6)
        except Exception as e:                              # noqa: BLE001
            errs.append((i, repr(e)))
    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for i in range(4):
        assert len(got[i]) == len(want[i])
        for j, (a, b) in enumerate(zip(got[i], want[i])):
            assert np.array_equal(a, b), "thread %d frame %d" % (i, j)
