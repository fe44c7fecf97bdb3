#!/usr/bin/env python3
"""Kernel time of single 8-row bands of the 1080p dragon frame (one tile row = 240 tiles) on an otherwise idle GPU:
the chain latency of the heaviest tiles without any contention."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd as R
s = R.Scene(R.assets.mesh_path("dragon_vis.ply")); s.bvh_update()
cam, lights, n = R.benchmark_frame(0)
W, H = 1920, 1080
for k in (0, 40, 50, 60, 67, 75, 85, 100, 110):
    o = R.default_opts(W, H, band_rows=8, band_index=k, band_count=135, compact_rows=1)
    s.render(9, cam, lights, n, o)
    ms = min(s.render(9, cam, lights, n, o)[2].kernel_ms for _ in range(5))
    st = s.render(9, cam, lights, n, o)[2]
    print("tile row %3d (y=%4d): %.3f ms, rays %d" % (k, k * 8, ms, st.normal_rays + st.shadow_rays), flush=True)
for bc in (135, 27, 9, 3, 1):
    o = R.default_opts(W, H, band_rows=8, band_index=bc // 2, band_count=bc, compact_rows=1)
    s.render(9, cam, lights, n, o)
    ms = min(s.render(9, cam, lights, n, o)[2].kernel_ms for _ in range(5))
    print("every %d-th tile row (1/%d of the frame): %.3f ms" % (bc, bc, ms), flush=True)
