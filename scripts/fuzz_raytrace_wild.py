#!/usr/bin/env python3
"""Randomised parity sweep of the raytracer beyond the benchmark orbit (not part of the test suite): random soups and the
shipped meshes, cameras anywhere (inside the model, grazing, far away), one or two lights anywhere (inside the geometry too),
random option sets (depth 1-4, shadows / reflections on or off, 1 or 4 samples, odd frame sizes, bands, every walk and wave
build) -- production frames, float frames and counters against the oracle."""
import argparse, json, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd as R
from oracle import oracle_ctypes as O

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=40)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
O.build()
tmp = tempfile.mkdtemp()
bad = 0
scenes = {}


def scene_pair(rng, it):
    pick = int(rng.integers(0, 6))
    if pick < 3:
        name = ["dragon_vis.ply", "chessboard.tri", "legocar.3ds"][pick]
        if name not in scenes:
            d = R.Scene(R.assets.mesh_path(name)); d.bvh_create("device")
            o = O.Scene(R.assets.oracle_path(name)); o.bvh_ensure(os.path.join(R.assets.cache_dir(), name + ".oracle.bvh"))
            scenes[name] = (d, o)
        return name, scenes[name]
    n_tri = int(rng.choice([1, 5, 40, 500, 4000]))
    size = float(rng.choice([0.05, 0.3, 1.5]))
    snap = [None, 0.25, 0.03125][int(rng.integers(0, 3))]
    v = rng.uniform(-1, 1, (n_tri, 1, 3)) + rng.uniform(-size, size, (n_tri, 3, 3))
    if snap: v = np.round(v / snap) * snap
    verts = v.reshape(-1, 3); faces = np.arange(verts.shape[0]).reshape(-1, 3)
    p = os.path.join(tmp, "w%d.ply" % it)
    with open(p, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nelement face %d\nend_header\n" % (len(verts), len(faces)))
        for q in verts: f.write("%r %r %r %d\n" % (float(q[0]), float(q[1]), float(q[2]), int(rng.integers(0, 256))))
        for t in faces: f.write("3 %d %d %d %d %d %d\n" % (t[0], t[1], t[2], *[int(x) for x in rng.integers(0, 256, 3)]))
    d = R.Scene(p)
    if not np.isfinite(d.arrays()["vertex_pos"]).all():
        return None, None
    d.bvh_create("device")
    o = O.Scene(p); o.bvh_build()
    return "soup%d(%d)" % (it, n_tri), (d, o)


for it in range(args.n):
    rng = np.random.default_rng(args.seed * 104729 + it)
    name, pair = scene_pair(rng, it)
    if pair is None:
        continue
    d, o = pair
    for trial in range(3):
        eye = (rng.uniform(-1, 1, 3) * float(rng.choice([0.05, 0.5, 1.5, 6.0]))).astype(np.float32)
        look = rng.uniform(-0.4, 0.4, 3).astype(np.float32)
        cam, ocam = R.camera(eye, look), O.camera(eye, look)
        nl = int(rng.integers(1, 3))
        lights, ol = (R.Light * 2)(), (O.Light * 2)()
        for i in range(nl):
            lp = (rng.uniform(-1, 1, 3) * float(rng.choice([0.2, 2.0, 8.0]))).astype(np.float32)
            lights[i] = R.light(lp, cam); ol[i] = O.light(lp, ocam)
        W, H = [(320, 240), (333, 217), (96, 64), (641, 97)][int(rng.integers(0, 4))]
        kw = dict(max_ray_depth=int(rng.integers(1, 5)), use_shadows=int(rng.integers(0, 2)), use_reflections=int(rng.integers(0, 2)))
        tune = [{}, {"bpc": 3}, {"bpc": 4}, {"reforder": 1}, {"exact": 1}, {"xmin": 16, "rmin": 32}][int(rng.integers(0, 6))]
        mode = 10 if rng.random() < 0.2 else 9
        stats = int(rng.random() < 0.3)
        img, f32, st = d.render(mode, cam, lights, nl, R.default_opts(W, H, tune=tune, collect_stats=stats, **kw), want_f32=True)
        oi, of32, ost = o.render(mode, ocam, ol, nl, O.default_opts(W, H, threads=os.cpu_count() or 1, **kw), want_f32=True)
        diff = int((img != oi).sum()) + int((f32 != of32).sum()) + int(st.normal_rays != ost.normal_rays) + int(st.shadow_rays != ost.shadow_rays)
        if stats:
            diff += int(st.node_pops != ost.node_pops) + int(st.tri_tests != ost.tri_tests) + int(st.plane_pass != ost.plane_pass)
        if diff:
            bad += 1
            print("case %d %s trial %d mode %d %dx%d %s tune %s stats %d eye %s: %d differences" % (it, name, trial, mode, W, H, kw, json.dumps(tune), stats, eye.tolist(), diff), flush=True)
        elif os.environ.get("FUZZ_VERBOSE"):
            print("case %d %s trial %d ok" % (it, name, trial), flush=True)
print("raytrace wild fuzz: %d cases, %d bad frames" % (args.n, bad))
