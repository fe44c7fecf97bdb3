#!/usr/bin/env python3
"""Phase breakdown of the longest-running waves of one raytraced frame (counting build)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd as R
s = R.Scene(R.assets.mesh_path("dragon_vis.ply")); s.bvh_update()
cam, lights, n = R.benchmark_frame(0)
L = R.lib(); L.mi355i_fetch_wave_profiles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
names = ["cyc_total", "cyc_refill", "cyc_trans", "cyc_inner", "cyc_leaf", "it_refill", "ln_refill", "it_trans", "ln_trans",
         "it_inner", "ln_inner", "it_leaf", "ln_leaf", "one", "lds", "loops"]
for label, kw in (("full frame", {}), ("tile row 67", dict(band_rows=8, band_index=67, band_count=135, compact_rows=1))):
    o = R.default_opts(1920, 1080, collect_stats=1, **kw)
    s.render(9, cam, lights, n, o)
    _, _, st = s.render(9, cam, lights, n, o)
    buf = np.zeros((8192, 16), np.uint64)
    nw = L.mi355i_fetch_wave_profiles(s.context(), buf.ctypes.data, 8192)
    w = buf[:nw].astype(np.float64)
    order = np.argsort(-w[:, 0])
    print("==", label, "kernel_ms %.3f" % st.kernel_ms, "waves", nw, "avg cyc_total %.0f" % w[:, 0].mean())
    for i in order[:4]:
        d = dict(zip(names, w[i]))
        print("  wave %4d: total %.2fM cyc | refill %.2fM trans %.2fM inner %.2fM leaf %.2fM | loops %d it_inner %d it_leaf %d it_trans %d it_refill %d | lanes/inner-it %.1f lanes/leaf-it %.1f lanes/trans %.1f"
              % (i, d["cyc_total"] / 1e6, d["cyc_refill"] / 1e6, d["cyc_trans"] / 1e6, d["cyc_inner"] / 1e6, d["cyc_leaf"] / 1e6,
                 d["loops"], d["it_inner"], d["it_leaf"], d["it_trans"], d["it_refill"],
                 d["ln_inner"] / max(d["it_inner"], 1), d["ln_leaf"] / max(d["it_leaf"], 1), d["ln_trans"] / max(d["it_trans"], 1)))
