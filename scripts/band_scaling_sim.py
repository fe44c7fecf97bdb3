#!/usr/bin/env python3
"""One GPU plays every rank of an N-GPU step in turn (dragon 1080p, mode 9): a step of 8 N frames split as interleaved
bands (band_rows scanlines) or as whole frames; prints the slowest rank's launch time per N -- the step time an N-GPU
run would see before the gather."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import renderer_amd as R

ap = argparse.ArgumentParser()
ap.add_argument("--band-rows", default="15,8,16,24")
ap.add_argument("--gpus", default="1,2,4,8")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--reps", type=int, default=12)
args = ap.parse_args()
dev = torch.device("cuda", 0)
s = R.Scene(R.assets.mesh_path("dragon_vis.ply")); s.bvh_update()
W, H = args.width, args.height
stream = torch.cuda.current_stream(dev)
cams = [R.benchmark_frame(k) for k in range(200)]


def launch_ms(frames, opts, rows):
    bufs = [torch.zeros((rows, W), dtype=torch.int32, device=dev) for _ in frames]
    def go(shift):
        fs = [(f + shift) % 200 for f in frames]
        s.render_batch_device(9, [cams[f][0] for f in fs], [cams[f][1] for f in fs], 1, opts, [b.data_ptr() for b in bufs], W * 4, None, stream.cuda_stream)
    go(0); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.reps): go(i * len(frames))
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / args.reps * 1e3


for N in [int(x) for x in args.gpus.split(",")]:
    B = 8 * N
    step = list(range(B))
    whole = max(launch_ms(step[r::N], R.default_opts(W, H), H) for r in range(N))
    line = {"gpus": N, "frames_per_step": B, "whole_frames_ms": round(whole, 3)}
    if N > 1:
        for br in [int(x) for x in args.band_rows.split(",")]:
            worst = 0.0
            for r in range(N):
                o = R.default_opts(W, H, band_rows=br, band_index=r, band_count=N, compact_rows=1)
                rows = sum(1 for y in range(H) if (y // br) % N == r)
                worst = max(worst, launch_ms(step, o, rows))
            line["bands_%d_rows_ms" % br] = round(worst, 3)
    print(json.dumps(line), flush=True)
