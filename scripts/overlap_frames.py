#!/usr/bin/env python3
"""Throughput with two frames in flight (two contexts, two streams) vs one, dragon 1080p mode 9."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import renderer_amd as R
dev = torch.device("cuda", 0)
W, H, K = 1920, 1080, 200
cams = [R.benchmark_frame(k) for k in range(K)]
def run(n_ctx, tune):
    scenes = [R.Scene(R.assets.mesh_path("dragon_vis.ply")) for _ in range(n_ctx)]
    for s in scenes: s.bvh_update()
    streams = [torch.cuda.Stream(dev) for _ in range(n_ctx)]
    bufs = [torch.zeros((H, W), dtype=torch.int32, device=dev) for _ in range(n_ctx)]
    o = R.default_opts(W, H, tune=tune)
    for k in range(20):
        i = k % n_ctx
        scenes[i].render_device(9, *cams[k], o, bufs[i].data_ptr(), W * 4, 0, streams[i].cuda_stream)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(K):
        i = k % n_ctx
        scenes[i].render_device(9, *cams[k], o, bufs[i].data_ptr(), W * 4, 0, streams[i].cuda_stream)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    return K / dt
for n_ctx, tune in ((1, {}), (2, {}), (2, {"bpc": 1}), (3, {"bpc": 1}), (3, {}), (4, {"bpc": 1})):
    print("frames in flight %d tune %s: %.1f frames/s" % (n_ctx, json.dumps(tune), run(n_ctx, tune)), flush=True)
