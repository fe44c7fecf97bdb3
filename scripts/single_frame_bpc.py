#!/usr/bin/env python3
"""Kernel time of ONE raytraced 1080p frame (dragon, orbit frames) for the blocks-per-CU / wave builds: mi355_stats::kernel_ms."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd as R
s = R.Scene(R.assets.mesh_path("dragon_vis.ply")); s.bvh_create()
for label, t in (("default", {}), ("bpc2", dict(bpc=2)), ("bpc3", dict(bpc=3)), ("bpc4", dict(bpc=4)), ("xmin32", dict(xmin=32)), ("xmin16", dict(xmin=16)), ("xmin8", dict(xmin=8)), ("xmin1", dict(xmin=1)), ("xmin16 bpc4", dict(xmin=16, bpc=4)), ("xmin1 bpc4", dict(xmin=1, bpc=4))):
    ms = []
    for k in list(range(0, 200, 10)) * 2:
        cam, lights, n = R.benchmark_frame(k)
        _, _, st = s.render(9, cam, lights, n, R.default_opts(1920, 1080, tune=R.tune(**t)))
        ms.append(st.kernel_ms)
    ms = np.array(ms[20:])
    print("%-12s kernel_ms mean %.3f min %.3f max %.3f" % (label, ms.mean(), ms.min(), ms.max()))
