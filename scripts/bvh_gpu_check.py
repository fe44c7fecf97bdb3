#!/usr/bin/env python3
"""GPU SAH builder vs the host builder: same bytes? how long?"""
import hashlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd as R
for mesh in ("dragon_vis.ply", "statue.ply", "chessboard.tri"):
    s = R.Scene(R.assets.mesh_path(mesh))
    t0 = time.perf_counter(); s.bvh_create("host"); t_cpu = time.perf_counter() - t0
    nodes, idx = s.bvh_arrays()
    nodes, idx = nodes.copy(), idx.copy()
    s.context()
    g = s.build_bvh_device()                      # warm (allocations, code load)
    t0 = time.perf_counter(); g = s.build_bvh_device(); t_gpu = time.perf_counter() - t0
    import ctypes as C
    tm = (C.c_double * 4)(); R.lib().mi355i_bvh_last_times(tm)
    print("   GPU builder: setup %.2f ms, kernels of %d levels + flatten + streams %.2f ms, tree to the caller %.2f ms, install in the context %.2f ms" % (tm[0], g[2] + 1, tm[1], tm[2], tm[3]))
    same_nodes = g[0].shape == nodes.shape and bool((g[0] == nodes).all())
    same_idx = bool((g[1] == idx).all())
    blob = np.array([nodes.shape[0], idx.shape[0]], np.uint32).tobytes() + g[0].tobytes() + g[1].tobytes()
    print("%-16s nodes %d/%d depth %d | nodes identical %s, triangle list identical %s | .bvh sha256 %s | host builder %.1f ms, GPU builder %.2f ms"
          % (mesh, g[0].shape[0], nodes.shape[0], g[2], same_nodes, same_idx, hashlib.sha256(blob).hexdigest()[:16], t_cpu * 1e3, t_gpu * 1e3), flush=True)
    if not same_nodes and g[0].shape == nodes.shape:
        bad = np.argwhere((g[0] != nodes).any(axis=1))[:5, 0]
        for b in bad: print("   node", b, "gpu", g[0][b], "host", nodes[b])
