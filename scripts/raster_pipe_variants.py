#!/usr/bin/env python3
"""Frame-by-frame raster fps (chessboard 1080p, one caller stream, one output buffer) for the three ways the library can run
consecutive frames: overlapped (default), the ordered pipeline (tune flag 64), everything on the caller's stream (flag 32).
Prints the GPU-bound rate (enqueue N frames, then wait) and the host's enqueue time per frame."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import renderer_amd as R
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream(dev)
W, H, N = 1920, 1080, 400
s = R.Scene(R.assets.mesh_path("chessboard.tri"))
cams = [R.benchmark_frame(k % 200) for k in range(N)]
s.shadowmap_render(0, cams[0][1][0])
buf = torch.zeros((H, W), dtype=torch.int32, device=dev)
out = {}
only = sys.argv[1:]
for name, tn in (("overlapped", R.tune()), ("ordered", R.tune(pipeordered=1)), ("one_stream", R.tune(nopipe=1))):
    if only and name not in only: continue
    for mode in (4, 6, 8):
        o = R.default_opts(W, H, tune=tn)
        best, host = 0.0, 0.0
        for rep in range(3):
            for k in range(10): s.render_device(mode, *cams[k], o, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
            torch.cuda.synchronize(dev); t = time.perf_counter()
            for k in range(N): s.render_device(mode, *cams[k], o, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
            t1 = time.perf_counter(); torch.cuda.synchronize(dev); t2 = time.perf_counter()
            if N / (t2 - t) > best: best, host = N / (t2 - t), (t1 - t) / N * 1e6
        out["%s_mode%d" % (name, mode)] = {"fps": round(best, 1), "host_us_per_frame": round(host, 1)}
print(json.dumps(out))
