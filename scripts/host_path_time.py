#!/usr/bin/env python3
"""Frame time through the host-buffer entry point mi355_render (kernel + 8.3 MB D2H copy per 1080p frame)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd as R
s = R.Scene(R.assets.mesh_path("dragon_vis.ply")); s.bvh_update()
o = R.default_opts(1920, 1080)
cams = [R.benchmark_frame(k) for k in range(50)]
for c in cams[:5]: s.render(9, *c, o)
t0 = time.perf_counter(); ks = 0.0
for c in cams:
    ks += s.render(9, *c, o)[2].kernel_ms
dt = (time.perf_counter() - t0) / len(cams)
print("mi355_render (host buffers), dragon 1080p: %.3f ms/frame wall, %.3f ms of it kernel" % (dt * 1e3, ks / len(cams)))
