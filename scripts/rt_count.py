#!/usr/bin/env python3
"""Counters of the production walk loop (variant built with -DRT_COUNT=1 -DRT_DEFER=0: the in-step loop of the batch builds on single frames): iterations of a wave and lanes per phase, single frames
of the bench workload on the four-wave build.  MI355_RENDER_SO must point at variant_count.so."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import renderer_amd as R
names = ["iterations", "it_inner", "lanes_inner", "it_tri", "lanes_tri", "it_cand", "lanes_cand", "transitions", "lanes_trans", "leaf_entries", "lanes_idle", "steal_events", "steals",
         "it_one_record", "sum_distinct_records", "it_two_records", "-", "-", "it_3_4_records", "lanes_one_record"]
for mesh, depth in (("dragon_vis.ply", 3), ("statue.ply", 1)):
    s = R.Scene(R.assets.mesh_path(mesh)); s.bvh_create()
    for noshare in (0, 1):
        o = R.default_opts(1920, 1080, max_ray_depth=depth, tune=R.tune(bpc=4, noshare=noshare))
        tot = np.zeros(20)
        for k in (0, 50, 100, 150):
            cam, lights, n = R.benchmark_frame(k)
            _, _, st = s.render(9, cam, lights, n, o)
            out = (C.c_ulonglong * 20)()
            R.lib().mi355i_fetch_profile.argtypes = [C.c_void_p, C.c_void_p]
            assert R.lib().mi355i_fetch_profile(s.context(), out) == 0
            tot += np.array([int(out[i]) for i in range(20)])
        d = {n: int(v / 4) for n, v in zip(names, tot)}
        d.update(mesh=mesh, noshare=noshare, rays=int(st.normal_rays + st.shadow_rays), kernel_ms=round(st.kernel_ms, 4))
        print(json.dumps(d), flush=True)
