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
         "it_inner", "ln_inner", "it_leaf", "ln_leaf", "one", "cyc_wait", "loops"]
for label, kw in (("full frame, reference order", {}), ("full frame, ordered walk", dict(tune=dict(profordered=1))),
                  ("tile row 67, ordered walk", dict(band_rows=8, band_index=67, band_count=135, compact_rows=1, tune=dict(profordered=1)))):
    o = R.default_opts(1920, 1080, collect_stats=1, **kw)
    s.render(9, cam, lights, n, o)
    _, _, st = s.render(9, cam, lights, n, o)
    buf = np.zeros((8192, 16), np.uint64)
    nw = L.mi355i_fetch_wave_profiles(s.context(), buf.ctypes.data, 8192)
    w = buf[:nw].astype(np.float64)
    order = np.argsort(-w[:, 0])
    print("==", label, "kernel_ms %.3f" % st.kernel_ms, "waves", nw, "avg cyc_total %.0f" % w[:, 0].mean(),
          "| box tests %d tri tests %d plane pass %d" % (st.node_pops, st.tri_tests, st.plane_pass))
    prof = (C.c_uint64 * 20)()
    L.mi355i_fetch_profile.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.mi355i_fetch_profile(s.context(), prof)
    print("   all waves: cycles total %.0fM wait %.0fM inner %.0fM leaf+judge %.0fM trans %.0fM refill %.0fM | lanes sent to the exact box test %d"
          % (prof[0] / 1e6, prof[14] / 1e6, prof[3] / 1e6, prof[4] / 1e6, prof[2] / 1e6, prof[1] / 1e6, prof[15]))
    for i in order[:4]:
        d = dict(zip(names, w[i]))
        print("  wave %4d: total %.2fM cyc | wait %.2fM refill %.2fM trans %.2fM inner %.2fM leaf %.2fM | loops %d it_inner %d it_leaf %d it_trans %d it_refill %d | lanes/inner-it %.1f lanes/leaf-it %.1f lanes/trans %.1f"
              % (i, d["cyc_total"] / 1e6, d["cyc_wait"] / 1e6, d["cyc_refill"] / 1e6, d["cyc_trans"] / 1e6, d["cyc_inner"] / 1e6, d["cyc_leaf"] / 1e6,
                 d["loops"], d["it_inner"], d["it_leaf"], d["it_trans"], d["it_refill"],
                 d["ln_inner"] / max(d["it_inner"], 1), d["ln_leaf"] / max(d["it_leaf"], 1), d["ln_trans"] / max(d["it_trans"], 1)))
