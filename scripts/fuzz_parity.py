#!/usr/bin/env python3
"""Randomised parity sweep (not part of the test suite): GPU BVH builder vs host builder byte for byte, and traced
frames of the production kernel vs the oracle, over random triangle soups (snapped to grids to force ties)."""
import argparse, json, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd as R
from oracle import oracle_ctypes as O

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=60)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--tune", default="{}", help="JSON dict of renderer_amd.tune() knobs, e.g. '{\"bpc\": 3}' to force the three-wave build")
args = ap.parse_args()
O.build()
tmp = tempfile.mkdtemp()
bad = 0
for it in range(args.n):
    rng = np.random.default_rng(args.seed * 100003 + it)
    n_tri = int(rng.choice([1, 3, 4, 5, 9, 40, 300, 1500, 6000]))
    snap = [None, None, 0.5, 0.125, 0.03125][int(rng.integers(0, 5))]
    stretch = np.array([1.0, 1.0, 1.0]) if rng.random() < 0.6 else rng.choice([0.0, 0.05, 1.0, 8.0], 3)
    if not stretch.any(): stretch[0] = 1.0
    c = rng.uniform(-1, 1, (n_tri, 1, 3))
    v = (c + rng.uniform(-0.15, 0.15, (n_tri, 3, 3))) * stretch
    if snap: v = np.round(v / snap) * snap
    if rng.random() < 0.3: v = np.concatenate([v, v[: max(1, n_tri // 3)]])          # duplicates
    verts = v.reshape(-1, 3); faces = np.arange(verts.shape[0]).reshape(-1, 3)
    cols = rng.integers(20, 255, (faces.shape[0], 3))
    if os.environ.get("FUZZ_VERBOSE"):
        print("case %d: tris %d snap %s stretch %s" % (it, n_tri, snap, stretch), flush=True)
    p = os.path.join(tmp, "f%d.ply" % it)
    with open(p, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nelement face %d\nend_header\n" % (len(verts), len(faces)))
        for q in verts: f.write("%r %r %r 150\n" % (float(q[0]), float(q[1]), float(q[2])))
        for t, cc in zip(faces, cols): f.write("3 %d %d %d %d %d %d\n" % (t[0], t[1], t[2], cc[0], cc[1], cc[2]))
    if not np.isfinite(R.Scene(p).arrays()["vertex_pos"]).all():
        continue                       # collapsed to a point by the snapping: the loader's rescale makes NaNs of it
    try:
        d = R.Scene(p); d.bvh_create("device")
        h = R.Scene(p); h.bvh_create("host")
    except R.Mi355Error as e:
        print("case %d (%d tris): builder error: %s" % (it, faces.shape[0], e)); bad += 1; continue
    dn, di = d.bvh_arrays(); hn, hi = h.bvh_arrays()
    same_tree = dn.shape == hn.shape and bool((dn == hn).all()) and bool((di == hi).all())
    o = O.Scene(p); o.bvh_build()
    diff = 0
    for frame in (int(rng.integers(0, 200)), int(rng.integers(0, 200))):
        cam, lights, n = R.benchmark_frame(frame); ocam, ol, on = O.benchmark_frame(frame)
        W, H = 320, 240
        mode = 10 if rng.random() < 0.25 else 9
        img, f32, st = d.render(mode, cam, lights, n, R.default_opts(W, H, tune=json.loads(args.tune)), want_f32=True)
        oi, of32, ost = o.render(mode, ocam, ol, on, O.default_opts(W, H, threads=os.cpu_count() or 1), want_f32=True)
        diff += int((img != oi).sum()) + int((f32 != of32).sum()) + int(st.normal_rays != ost.normal_rays) + int(st.shadow_rays != ost.shadow_rays)
    ok = same_tree and diff == 0
    if not ok:
        bad += 1
        print("case %d: tris %d snap %s stretch %s: same tree %s, frame differences %d, ordered walk %d" % (it, faces.shape[0], snap, stretch, same_tree, diff, d.walk_info()[0]))
print("fuzz: %d cases, %d bad" % (args.n, bad))
