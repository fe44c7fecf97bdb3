#!/usr/bin/env python3
"""Wall time of mi355_build_bvh (whole call) and its four parts, repeated, per mesh."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd as R
for mesh in ("dragon_vis.ply", "statue.ply", "chessboard.tri", "legocar.3ds"):
    s = R.Scene(R.assets.mesh_path(mesh)); s.context()
    for rep in range(6):
        t0 = time.perf_counter(); s.build_bvh_device(); wall = (time.perf_counter() - t0) * 1e3
        tm = (C.c_double * 4)(); R.lib().mi355i_bvh_last_times(tm)
        print("%-16s rep %d: wall %.2f ms = setup %.2f + kernels %.2f + tree to caller %.2f + install %.2f" % (mesh, rep, wall, tm[0], tm[1], tm[2], tm[3]), flush=True)
