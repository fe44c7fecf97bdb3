#!/usr/bin/env python3
"""Randomised parity sweep of the TILED rasterizer's kernel bodies on the CPU (tests/emu: rs_core.h compiled for the host)
against the oracle: random triangle soups (snapped to grids to force equal depths and shared edges, stretched, huge and
sliver triangles, cameras inside the cloud so that triangles straddle the near plane and the screen edges), modes 4-8,
pixels and the tris_drawn / spans / ztests counters, odd frame sizes, band sharding, small bins (overflow must be reported)."""
import argparse, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import renderer_amd as R
from oracle import oracle_ctypes as O
from emu import emu

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=40)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--first", type=int, default=0)
args = ap.parse_args()
O.build()
tmp = tempfile.mkdtemp()
bad = frames = 0
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
    streams = emu.scene_streams(d)
    W, H = [(320, 240), (333, 217), (64, 48), (1920, 1080), (17, 5), (1, 1), (2500, 33)][int(rng.integers(0, 7))]
    for trial in range(2):
        eye = (rng.uniform(-1, 1, 3) * float(rng.choice([0.3, 1.0, 3.0]))).astype(np.float32)
        look = (rng.uniform(-0.3, 0.3, 3)).astype(np.float32)
        cam = R.camera(eye, look); ocam = O.camera(eye, look)
        lp = (rng.uniform(-3, 3, 3)).astype(np.float32)
        lights = (R.Light * 2)(R.light(lp, cam)); ol = (O.Light * 2)(O.light(lp, ocam))
        maps = [o.shadowmap(ol[0])]
        for mode in (4, 5, 6, 7, 8):
            band = None
            if rng.random() < 0.25:
                band = (int(rng.choice([1, 3, 8, 15, 16])), int(rng.integers(0, 3)), 3, int(rng.integers(0, 2)))
            ho = R.default_opts(W, H, collect_stats=1); oo = O.default_opts(W, H, threads=1)
            if band:
                ho.band_rows, ho.band_index, ho.band_count, ho.compact_rows = band
                oo.band_rows, oo.band_index, oo.band_count = band[:3]
            cap = 0 if rng.random() < 0.85 else int(rng.integers(1, 400))
            outs, st, over = emu.render(d, mode, [cam], [lights], 1, ho, maps if mode in (7, 8) else None, bins_cap=cap, streams=streams)
            oi, _, ost = o.render(mode, ocam, ol, 1, oo, shadow_maps=maps if mode in (7, 8) else None)
            frames += 1
            if over:
                continue                       # bins too small: reported, the caller draws again with larger ones
            img = outs[0]
            if band:
                sel = np.array([y for y in range(H) if (y // band[0]) % band[2] == band[1]], int)
                ref = oi[sel] if band[3] else oi
                if not band[3]:
                    mask = np.zeros(H, bool); mask[sel] = True
                    img = img.copy(); img[~mask] = 0          # foreign rows are the caller's (the launcher clears them)
                    ref = ref.copy(); ref[~mask] = 0
            else:
                ref = oi
            diff = int((img != ref).sum())
            cnt_ok = band is not None or (st["tris_drawn"], st["spans"], st["ztests"]) == (ost.tris_drawn, ost.spans, ost.ztests)
            if diff or not cnt_ok:
                bad += 1
                print("case %d (tris %d snap %s size %s) trial %d mode %d %dx%d band %s: %d pixels differ, counters %s vs %s" % (
                    it, faces.shape[0], snap, size, trial, mode, W, H, band, diff, st, (ost.tris_drawn, ost.spans, ost.ztests)), flush=True)
print("raster emu fuzz: cases %d..%d, %d frames, %d bad" % (args.first, args.n, frames, bad))
