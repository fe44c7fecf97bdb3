#!/usr/bin/env python3
"""sync_host_fps (mi355_render into frame memory of the library's, dragon 1080p mode 9) for the MI355_FILL_FIRST given in the environment:
the number of waves of k_raytrace that start with the background of a zero-copy frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import renderer_amd as R
W, H = 1920, 1080
s = R.Scene(R.assets.mesh_path("dragon_vis.ply"))
s.bvh_create()
buf = R.host_array((H, W))
cams = [R.benchmark_frame(k) for k in range(200)]
o = R.default_opts(W, H)
for k in range(10):
    s.render_into(9, *cams[k], o, buf)
t = time.perf_counter(); kms = 0.0
for k in range(200):
    kms += s.render_into(9, *cams[k], o, buf).kernel_ms
dt = time.perf_counter() - t
print("fill_first %s: %.1f frames/s, kernel %.4f ms" % (os.environ.get("MI355_FILL_FIRST", "default"), 200 / dt, kms / 200), flush=True)
R.host_array_free(buf)
