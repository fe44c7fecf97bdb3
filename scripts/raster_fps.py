#!/usr/bin/env python3
"""Frames per second of the raster modes (chessboard 1080p), single frames and batches of 8, and the BVH build time."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import renderer_amd as R
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream(dev)
W, H = 1920, 1080
s = R.Scene(R.assets.mesh_path("chessboard.tri"))
cams = [R.benchmark_frame(k) for k in range(200)]
s.shadowmap_render(0, cams[0][1][0])
buf = torch.zeros((H, W), dtype=torch.int32, device=dev)
out = {}
for mode in (2, 4, 6, 8):
    o = R.default_opts(W, H)
    for k in range(5): s.render_device(mode, *cams[k], o, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
    torch.cuda.synchronize(dev); t = time.perf_counter()
    for k in range(200): s.render_device(mode, *cams[k], o, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
    torch.cuda.synchronize(dev); out["mode%d_fps" % mode] = round(200 / (time.perf_counter() - t), 1)
bufs = [torch.zeros((H, W), dtype=torch.int32, device=dev) for _ in range(8)]
for mode in (6, 8):
    o = R.default_opts(W, H)
    def step(i):
        fs = [(8 * i + j) % 200 for j in range(8)]
        s.render_batch_device(mode, [cams[f][0] for f in fs], [cams[f][1] for f in fs], 1, o, [b.data_ptr() for b in bufs], W * 4, None, stream.cuda_stream)
    for i in range(3): step(i)
    torch.cuda.synchronize(dev); t = time.perf_counter()
    for i in range(25): step(i)
    torch.cuda.synchronize(dev); out["mode%d_batch8_fps" % mode] = round(200 / (time.perf_counter() - t), 1)
d = R.Scene(R.assets.mesh_path("dragon_vis.ply")); d.context(); best = 1e9
for _ in range(5):
    t = time.perf_counter(); d.build_bvh_device(); best = min(best, (time.perf_counter() - t) * 1e3)
out["dragon_bvh_build_ms"] = round(best, 2)
print(json.dumps(out))
