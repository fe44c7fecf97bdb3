#!/usr/bin/env python3
"""One-off scale check of the frame size: 8K (7680x4320) and a 16384-wide strip, modes 9 / 6 / 2 against the oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd as R
from oracle import oracle_ctypes as O
O.build()
d = R.Scene(R.assets.mesh_path("dragon_vis.ply")); d.bvh_create("device")
o = O.Scene(R.assets.mesh_path("dragon_vis.ply")); o.bvh_ensure(os.path.join(R.assets.cache_dir(), "dragon_vis.ply.oracle.bvh"))
bad = 0
for (W, H) in ((7680, 4320), (16384, 64), (64, 16384)):
    for mode in (9, 6, 2):
        cam, lights, n = R.benchmark_frame(3); ocam, ol, on = O.benchmark_frame(3)
        t = time.time(); img, _, st = d.render(mode, cam, lights, n, R.default_opts(W, H)); tg = time.time() - t
        t = time.time(); oi, _, _ = o.render(mode, ocam, ol, on, O.default_opts(W, H, threads=(os.cpu_count() or 1) if mode >= 9 else 1)); to = time.time() - t
        diff = int((img != oi).sum())
        bad += diff != 0
        print("%dx%d mode %d: %d differences (GPU call %.2f s, kernel %.2f ms; oracle %.1f s)" % (W, H, mode, diff, tg, st.kernel_ms, to), flush=True)
print("big frame check:", "OK" if not bad else "FAILED")
