#!/usr/bin/env python3
"""Kernel time / Mrays/s of the BASELINE raytrace configs and their larger variants (one GPU)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd as R
cam, lights, n = R.benchmark_frame(0)
scenes = {}
for mesh, mode, W, H, depth, label in (
        ("statue.ply", 9, 1920, 1080, 1, "cfg3 statue primary+shadow 1080p"),
        ("statue.ply", 9, 1920, 1080, 3, "statue depth 3 1080p"),
        ("dragon_vis.ply", 9, 1920, 1080, 3, "cfg4 dragon depth 3 1080p"),
        ("dragon_vis.ply", 10, 1920, 1080, 3, "dragon 4spp AA 1080p"),
        ("dragon_vis.ply", 9, 3840, 2160, 3, "cfg5 dragon depth 3 4K (whole frame, 1 GPU)"),
        ("dragon_vis.ply", 10, 3840, 2160, 3, "dragon 4spp AA 4K"),
        ("chessboard.tri", 9, 1920, 1080, 3, "chessboard depth 3 1080p")):
    if mesh not in scenes:
        scenes[mesh] = R.Scene(R.assets.mesh_path(mesh)); scenes[mesh].bvh_update()
    s = scenes[mesh]
    o = R.default_opts(W, H, max_ray_depth=depth)
    s.render(mode, cam, lights, n, o)
    best = None
    for _ in range(5):
        st = s.render(mode, cam, lights, n, o)[2]
        if best is None or st.kernel_ms < best.kernel_ms: best = st
    rays = best.normal_rays + best.shadow_rays
    os_ = R.default_opts(W, H, max_ray_depth=depth, collect_stats=1)
    sc = s.render(mode, cam, lights, n, os_)[2].as_dict()
    B = 32 * sc["node_pops"] + 36 * sc["tri_tests"] + 48 * sc["plane_pass"] + 96 * sc["shaded_hits"] + 4 * W * H
    print(json.dumps({"config": label, "kernel_ms": round(best.kernel_ms, 3), "rays": rays, "Mrays_s": round(rays / best.kernel_ms / 1e3, 1),
                      "alg_GB": round(B / 1e9, 3), "frac_of_8TBs": round(B / (best.kernel_ms * 1e-3) / 8e12, 3)}), flush=True)
