#!/usr/bin/env python3
"""Lane utilisation of the ordered walk (counting build of it): lanes per box-section / triangle-section iteration, summed over
all waves of a frame, and how the loop iterations split."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd as R
L = R.lib(); L.mi355i_fetch_wave_profiles.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
names = ["cyc_total", "cyc_refill", "cyc_trans", "cyc_inner", "cyc_leaf", "it_refill", "ln_refill", "it_trans", "ln_trans",
         "it_inner", "ln_inner", "it_leaf", "ln_leaf", "one", "cyc_wait", "loops"]
for mesh in ("dragon_vis.ply", "chessboard.tri"):
    s = R.Scene(R.assets.mesh_path(mesh)); s.bvh_update()
    cam, lights, n = R.benchmark_frame(0)
    o = R.default_opts(1920, 1080, collect_stats=1, tune=dict(profordered=1))
    s.render(9, cam, lights, n, o)
    _, _, st = s.render(9, cam, lights, n, o)
    buf = np.zeros((8192, 16), np.uint64)
    nw = L.mi355i_fetch_wave_profiles(s.context(), buf.ctypes.data, 8192)
    w = buf[:nw].astype(np.float64)
    t = dict(zip(names, w.sum(axis=0)))
    print("%s: %d waves, loops %.0fk; box iterations %.0fk with %.1f lanes each; triangle iterations %.0fk with %.1f lanes each; "
          "transition phases %.0fk with %.1f lanes; refills %.0fk with %.1f lanes" % (mesh, nw, t["loops"] / 1e3, t["it_inner"] / 1e3,
          t["ln_inner"] / max(t["it_inner"], 1), t["it_leaf"] / 1e3, t["ln_leaf"] / max(t["it_leaf"], 1), t["it_trans"] / 1e3,
          t["ln_trans"] / max(t["it_trans"], 1), t["it_refill"] / 1e3, t["ln_refill"] / max(t["it_refill"], 1)))
    print("   cycles: total %.0fM = wait %.0fM + box %.0fM + triangle/judge %.0fM + transitions %.0fM + refill %.0fM" % (
        t["cyc_total"] / 1e6, t["cyc_wait"] / 1e6, t["cyc_inner"] / 1e6, t["cyc_leaf"] / 1e6, t["cyc_trans"] / 1e6, t["cyc_refill"] / 1e6))
    # distribution of active lanes: loops with k walking lanes is not recorded; the ratio of lane-iterations to 64 * loops is
    walking = (t["ln_inner"] + t["ln_leaf"]) / (64.0 * max(t["loops"], 1))
    print("   lanes doing a box or a triangle test per loop iteration: %.1f of 64 (%.0f %%)" % (walking * 64, walking * 100))
