#!/usr/bin/env python3
"""Thread scaling of the CPU oracle's raytracer on this box (picks the baseline's thread count)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_ctypes as O
from renderer_amd import assets
s = O.Scene(assets.mesh_path("dragon_vis.ply")); s.bvh_ensure(os.path.join(assets.cache_dir(), "dragon_vis.ply.oracle.bvh"))
cam, lights, n = O.benchmark_frame(0)
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for t in (1, 8, 16, 32, 64, 128, 256):
    o = O.default_opts(1920, 1080, threads=t)
    s.render(9, cam, lights, n, o)
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < 1.5:
        s.render(9, cam, lights, n, o); k += 1
    print("threads %3d: %.2f fps" % (t, k / (time.perf_counter() - t0)), flush=True)
