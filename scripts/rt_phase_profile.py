#!/usr/bin/env python3
"""Where the waves of the counting builds spend their cycles (STATS builds of k_raytrace: refill / transitions / walk phases and the wait for the next record),
dragon 1080p frame 0: the ordered walk (tune flag 8) and the reference-order walk."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import renderer_amd as R
s = R.Scene(R.assets.mesh_path("dragon_vis.ply")); s.bvh_create()
names = ["pc_total","pc_refill","pc_trans","pc_a","pc_b","it_refill","ln_refill","it_trans","ln_trans","it_a","ln_a","it_b","ln_b","waves","pc_wait"]
for tune in (dict(profordered=1), dict()):
    o = R.default_opts(1920, 1080, tune=R.tune(**tune), collect_stats=1)
    cam, lights, n = R.benchmark_frame(0)
    _, _, st = s.render(9, cam, lights, n, o)
    out = (C.c_ulonglong * 20)()
    R.lib().mi355i_fetch_profile.argtypes = [C.c_void_p, C.c_void_p]
    assert R.lib().mi355i_fetch_profile(s.context(), out) == 0
    d = {k: int(out[i]) for i, k in enumerate(names)}
    tot = d["pc_total"]
    print(tune, "kernel_ms", round(st.kernel_ms, 3), {k: round(d[k] / tot, 3) for k in ("pc_refill", "pc_trans", "pc_a", "pc_b", "pc_wait")}, "refills", d["it_refill"], "transitions", d["it_trans"], "waves", d["waves"])
