#!/usr/bin/env python3
"""Time to the first raytraced frame of a freshly loaded model: load, context (uploads), BVH build, first frame, next frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd as R
R.Scene(R.assets.mesh_path("chessboard.tri")).context()          # HIP runtime + library start-up, not counted
for mesh in ("dragon_vis.ply", "statue.ply", "legocar.3ds", "dragon_vis.ply"):
    p = R.assets.mesh_path(mesh)
    cam, lights, n = R.benchmark_frame(0)
    t0 = time.perf_counter(); s = R.Scene(p)
    t1 = time.perf_counter(); s.context()
    t2 = time.perf_counter(); s.build_bvh_device()
    t3 = time.perf_counter(); s.render(9, cam, lights, n, R.default_opts(1920, 1080))
    t4 = time.perf_counter(); s.render(9, cam, lights, n, R.default_opts(1920, 1080))
    t5 = time.perf_counter()
    print("%-16s load %.1f ms, context + uploads %.1f ms, BVH build %.1f ms, first frame %.1f ms, next frame %.1f ms -> first pixel after %.1f ms"
          % (mesh, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t4 - t0) * 1e3), flush=True)
