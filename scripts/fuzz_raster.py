#!/usr/bin/env python3
"""Randomised parity sweep of the raster modes (not part of the test suite): random triangle soups (snapped to grids to force
equal depths and shared edges, stretched, with huge and sliver triangles, cameras inside the cloud so that triangles straddle
the near plane and the screen edges) -- modes 1, 2, 4, 5, 6, 7, 8 against the oracle, pixel for pixel."""
import argparse, os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd as R
from oracle import oracle_ctypes as O

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=40)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--first", type=int, default=0, help="skip the cases before this one")
args = ap.parse_args()
O.build()
tmp = tempfile.mkdtemp()
bad = 0
for it in range(args.first, args.n):
    rng = np.random.default_rng(args.seed * 7919 + it)
    n_tri = int(rng.choice([1, 2, 7, 60, 400, 3000]))
    snap = [None, None, 0.25, 0.0625][int(rng.integers(0, 4))]
    size = float(rng.choice([0.05, 0.2, 0.8, 2.5]))
    c = rng.uniform(-1, 1, (n_tri, 1, 3))
    v = c + rng.uniform(-size, size, (n_tri, 3, 3))
    if rng.random() < 0.3: v[:, :, int(rng.integers(0, 3))] *= 0.02          # flat cloud
    if snap: v = np.round(v / snap) * snap
    if rng.random() < 0.3: v = np.concatenate([v, v[: max(1, n_tri // 2)]])   # exact duplicates: equal depth everywhere
    verts = v.reshape(-1, 3); faces = np.arange(verts.shape[0]).reshape(-1, 3)
    cols = rng.integers(0, 256, (faces.shape[0], 3))
    ao = rng.integers(0, 256, verts.shape[0])
    p = os.path.join(tmp, "r%d.ply" % it)
    with open(p, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nelement face %d\nend_header\n" % (len(verts), len(faces)))
        for q, a in zip(verts, ao): f.write("%r %r %r %d\n" % (float(q[0]), float(q[1]), float(q[2]), a))
        for t, cc in zip(faces, cols): f.write("3 %d %d %d %d %d %d\n" % (t[0], t[1], t[2], cc[0], cc[1], cc[2]))
    try:
        d = R.Scene(p)
    except R.Mi355Error:
        continue
    if not np.isfinite(d.arrays()["vertex_pos"]).all():
        continue
    o = O.Scene(p)
    W, H = [(320, 240), (333, 217), (64, 48)][int(rng.integers(0, 3))]
    for trial in range(2):
        # eye anywhere from inside the cloud to well outside, looking roughly at the centre
        eye = (rng.uniform(-1, 1, 3) * float(rng.choice([0.3, 1.0, 3.0]))).astype(np.float32)
        look = (rng.uniform(-0.3, 0.3, 3)).astype(np.float32)
        cam = R.camera(eye, look); ocam = O.camera(eye, look)
        lp = (rng.uniform(-3, 3, 3)).astype(np.float32)
        lights = (R.Light * 2)(R.light(lp, cam)); ol = (O.Light * 2)(O.light(lp, ocam))
        for mode in (1, 2, 4, 5, 6, 7, 8):
            if os.environ.get("FUZZ_VERBOSE"):
                print("case %d tris %d snap %s size %s trial %d mode %d %dx%d eye %s" % (it, faces.shape[0], snap, size, trial, mode, W, H, eye.tolist()), flush=True)
            maps = None
            if mode in (7, 8):
                maps = [o.shadowmap(ol[0])]
                d.shadowmap_render(0, lights[0])
            img, _, st = d.render(mode, cam, lights, 1, R.default_opts(W, H))
            oi, _, ost = o.render(mode, ocam, ol, 1, O.default_opts(W, H, threads=1), shadow_maps=maps)
            diff = int((img != oi).sum())
            if diff:
                bad += 1
                print("case %d (tris %d snap %s size %s) trial %d mode %d %dx%d: %d pixels differ" % (it, faces.shape[0], snap, size, trial, mode, W, H, diff), flush=True)
print("raster fuzz: %d cases, %d bad frames" % (args.n, bad))
