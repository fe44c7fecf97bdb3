#!/usr/bin/env python3
"""Tuning sweep of the raytrace kernel on one GPU: kernel time per variant + phase profile.

grid entries are renderer_amd.tune() keyword dicts
Prints one line per variant; optional --profile prints the in-kernel phase counters."""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd as R  # noqa: E402

PROF_NAMES = ["cyc_total", "cyc_refill", "cyc_trans", "cyc_inner", "cyc_leaf", "it_refill", "ln_refill", "it_trans",
              "ln_trans", "it_inner", "ln_inner", "it_leaf", "ln_leaf", "waves", "lds_visits"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mesh", default="dragon_vis.ply")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--grid", default="default")
    args = ap.parse_args()
    s = R.Scene(R.assets.mesh_path(args.mesh))
    s.bvh_update()
    cams = [R.benchmark_frame(k) for k in range(args.frames)]
    L = R.lib()
    L.mi355i_fetch_profile.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    ref_hash = None
    if args.grid == "default":
        grid = [dict()]
    elif args.grid == "wide":
        grid = [dict(trav=t, xmin=x, rmin=r, chunk=c) for t in (0, 1) for x in (4, 12, 24, 40) for r in (8, 24, 48)
                for c in (64, 256)]
    else:
        grid = json.loads(args.grid)
    for g in grid:
        o = R.default_opts(args.width, args.height, max_ray_depth=args.depth, tune=g)
        img, _, st = s.render(9, *cams[0], o)          # warm
        h = hashlib.sha256(img.tobytes()).hexdigest()[:12]
        if ref_hash is None:
            ref_hash = h
        ms = []
        rays = 0
        for k in range(args.frames):
            _, _, st = s.render(9, *cams[k], o)
            ms.append(st.kernel_ms)
            rays += st.normal_rays + st.shadow_rays
        line = {"variant": g, "ms_min": round(min(ms), 4), "ms_avg": round(sum(ms) / len(ms), 4),
                "Mrays_s": round(rays / (sum(ms) * 1e-3) / 1e6, 1), "same_pixels": h == ref_hash}
        if args.profile:
            o2 = R.default_opts(args.width, args.height, max_ray_depth=args.depth, tune=g, collect_stats=1)
            _, _, st2 = s.render(9, *cams[0], o2)
            prof = (C.c_uint64 * 20)()
            L.mi355i_fetch_profile(s.context(), prof)
            p = dict(zip(PROF_NAMES, [int(x) for x in prof]))
            w = max(p["waves"], 1)
            line["stats_ms"] = round(st2.kernel_ms, 4)
            line["cyc_per_wave"] = {k: round(p[k] / w) for k in PROF_NAMES[:5]}
            line["util"] = {ph: round(p["ln_" + ph] / max(p["it_" + ph], 1) / 64, 3) for ph in ("refill", "trans", "inner", "leaf")}
            t_start, t_dry, t_end, it_max = int(prof[16]), int(prof[17]), int(prof[18]), int(prof[19])
            line["dry_at_ms"] = round((t_dry - t_start) / 1e5, 4)
            line["end_at_ms"] = round((t_end - t_start) / 1e5, 4)
            line["max_wave_iters"] = it_max
            line["lds_visit_frac"] = round(p["lds_visits"] / max(st2.node_pops, 1), 3)
            line["iters_per_wave"] = {ph: round(p["it_" + ph] / w, 1) for ph in ("refill", "trans", "inner", "leaf")}
            line["counters"] = {k: st2.as_dict()[k] for k in ("node_pops", "tri_tests", "plane_pass", "shaded_hits")}
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
