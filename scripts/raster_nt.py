#!/usr/bin/env python3
"""Chessboard 1080p, modes 6 / 8, frame by frame: frames/s by threads per tile block (mi355_opts::tune[3])."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import renderer_amd as R
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream(dev)
W, H = 1920, 1080
s = R.Scene(R.assets.mesh_path("chessboard.tri"))
cams = [R.benchmark_frame(k) for k in range(200)]
s.shadowmap_render(0, cams[0][1][0])
buf = torch.zeros((H, W), dtype=torch.int32, device=dev)
ref = {}
for nt in [int(a) for a in sys.argv[1:]] or [256, 192, 320, 384, 128]:
    row = {"nt": nt}
    for mode in (6, 8):
        o = R.default_opts(W, H, tune=R.tune(rsnt=nt))
        for k in range(5): s.render_device(mode, *cams[k], o, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
        torch.cuda.synchronize(dev)
        best = 0.0
        for rep in range(3):
            t = time.perf_counter()
            for k in range(200): s.render_device(mode, *cams[k], o, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
            torch.cuda.synchronize(dev); best = max(best, 200 / (time.perf_counter() - t))
        row["mode%d_fps" % mode] = round(best, 1)
        px = np.array(s.render(mode, *cams[0], o)[0])
        row["mode%d_same" % mode] = bool(np.array_equal(ref.setdefault(mode, px), px))
    print(json.dumps(row), flush=True)
