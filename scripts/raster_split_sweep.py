#!/usr/bin/env python3
"""Heavy tiles in strips of rows (mi355_opts::tune[7], k_raster.hip: tile_order): frame-by-frame fps of the overlapped device path
and the kernel time of one synchronous frame, chessboard / dragon 1080p, for a few thresholds (-1 = never split)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import renderer_amd as R
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream(dev)
W, H, N = 1920, 1080, 400
for mesh in (sys.argv[1:] or ["chessboard.tri", "dragon_vis.ply"]):
    s = R.Scene(R.assets.mesh_path(mesh))
    cams = [R.benchmark_frame(k % 200) for k in range(N)]
    s.shadowmap_render(0, cams[0][1][0])
    buf = torch.zeros((H, W), dtype=torch.int32, device=dev)
    ref = {}
    for split in (-1, 192, 256, 384, 512, 768, 1024, 1536):
        row = {"mesh": mesh, "split": split}
        for mode in (6, 8):
            o = R.default_opts(W, H, tune=R.tune(rssplit=split))
            best = 0.0
            for rep in range(3):
                for k in range(10): s.render_device(mode, *cams[k], o, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
                torch.cuda.synchronize(dev); t = time.perf_counter()
                for k in range(N): s.render_device(mode, *cams[k], o, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
                torch.cuda.synchronize(dev)
                best = max(best, N / (time.perf_counter() - t))
            ms = [s.render(mode, *cams[k], o)[2].kernel_ms for k in (0, 50, 100, 150) for _ in range(3)]
            img = s.render(mode, *cams[7], o)[0]
            same = bool(np.array_equal(ref.setdefault(mode, img), img))
            row["mode%d" % mode] = {"fps": round(best, 1), "sync_kernel_ms": round(float(np.min(ms)), 4), "same_pixels": same}
        print(json.dumps(row), flush=True)
