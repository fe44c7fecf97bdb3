#!/usr/bin/env python3
"""Time of one 1024^2 shadow map (mi355_light_update, back to back on one stream; HIP events) per mesh."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import renderer_amd as R
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev)
for mesh in ("chessboard.tri", "dragon_vis.ply", "statue.ply"):
    s = R.Scene(R.assets.mesh_path(mesh))
    pos = [3.394, 3.394, 4.8]
    for _ in range(3): s.light_update(0, pos, 1024, stream.cuda_stream)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    e0.record(stream)
    for k in range(n):
        a = 0.785 + 0.01 * k
        s.light_update(0, [4.8 * np.cos(a), 4.8 * np.sin(a), 4.8], 1024, stream.cuda_stream)
    e1.record(stream)
    torch.cuda.synchronize(dev)
    s.fetch_stats()
    print("%s: %.1f us per 1024^2 map" % (mesh, e0.elapsed_time(e1) / n * 1e3), flush=True)
