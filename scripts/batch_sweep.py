#!/usr/bin/env python3
"""Throughput of batched raytrace launches (dragon 1080p, mode 9) over batch sizes and tuning knobs."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import renderer_amd as R
ap = argparse.ArgumentParser()
ap.add_argument("--grid", default='[[8, {}]]', help='JSON list of [frames_per_launch, tune dict]')
ap.add_argument("--frames", type=int, default=480)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
args = ap.parse_args()
dev = torch.device("cuda", 0)
s = R.Scene(R.assets.mesh_path("dragon_vis.ply")); s.bvh_update()
W, H = args.width, args.height
stream = torch.cuda.current_stream(dev)
cams = [R.benchmark_frame(k) for k in range(200)]
for B, tune in json.loads(args.grid):
    o = R.default_opts(W, H, tune=tune)
    bufs = [torch.zeros((H, W), dtype=torch.int32, device=dev) for _ in range(B)]
    def step(i):
        ks = [(i * B + j) % 200 for j in range(B)]
        if B == 1:
            s.render_device(9, *cams[ks[0]], o, bufs[0].data_ptr(), W * 4, 0, stream.cuda_stream)
        else:
            s.render_batch_device(9, [cams[k][0] for k in ks], [cams[k][1] for k in ks], 1, o, [b.data_ptr() for b in bufs], W * 4, None, stream.cuda_stream)
    for i in range(4): step(i)
    torch.cuda.synchronize(dev)
    n = max(1, args.frames // B)
    t0 = time.perf_counter()
    for i in range(n): step(i)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    print(json.dumps({"frames_per_launch": B, "tune": tune, "frames_per_s": round(n * B / dt, 1), "ms_per_launch": round(dt / n * 1e3, 3)}), flush=True)
