#!/usr/bin/env python3
"""What a single raytraced frame's time is made of, modelled on the CPU (no GPU needed): the oracle casts every pixel's rays
and prices each with a near-first walk with distance culling (oracle.cc: ordered_walk_cost, a model of the device's ordered
walk: one step per inner record visited, one per triangle tested).  A wave traces an 8x8 tile in lockstep generations -- camera
rays, then their shadow rays, then the reflected rays, ... -- so a tile's chain of dependent steps is the sum over generations of
the generation's LONGEST ray, and a frame alone on the GPU ends on its longest tile.  Printed per frame: that chain as built, and
under changes to the walk: (B) a hit's shadow ray walked beside its reflected ray (two lanes per pixel, generations still in
lockstep), (C) a four-wide tree (a step looks at the node's grandchildren, nearest first: walked, not estimated), (D) both,
(E) every shadow ray beside everything that follows its hit (the pixel's chain is then camera + reflections, with the shadow
rays hanging off it), and the deepest stack of postponed nodes the binary and the four-wide walk need.

    python scripts/chain_model.py [mesh] [frames, default 0,50,100,150]
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle_ctypes as O
import renderer_amd.assets as A

mesh = sys.argv[1] if len(sys.argv) > 1 else "dragon_vis.ply"
frames = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,50,100,150").split(",")]
W, H = 1920, 1080
s = O.Scene(A.mesh_path(mesh)); s.bvh_build()
print("%s, %dx%d, mode 9 (shadow rays, two reflection bounces); steps = inner records + triangles of a near-first walk" % (mesh, W, H))
print("frame | tiles with a hit | chain of the longest tile: as built | B shadow beside reflection | C four-wide tree | D both | E shadows off the chain | E with C | mean tile with a hit | stack binary / four-wide")
for f in frames:
    cam, lights, n = O.benchmark_frame(f)
    o = O.default_opts(W, H, threads=os.cpu_count() or 1)
    w, sp2 = s.chain_profile(cam, lights, n, o)
    w4, sp4 = s.chain_profile(cam, lights, n, o, quad=True)
    inner, tris = (w & 0xffff).astype(np.int64), (w >> 16).astype(np.int64)
    inner4, tris4 = (w4 & 0xffff).astype(np.int64), (w4 >> 16).astype(np.int64)
    def tiles(a):       # (H, W, 8) -> per 8x8 tile, per ray slot: the longest ray
        return a.reshape(H // 8, 8, W // 8, 8, 8).max(axis=(1, 3))
    def chains(steps):  # slots: 0 camera, 1 shadow, 2 reflection, 3 shadow, 4 reflection, 5 shadow
        m = tiles(steps)
        as_built = m[..., :6].sum(-1)
        beside = m[..., 0] + np.maximum(m[..., 1], m[..., 2]) + np.maximum(m[..., 3], m[..., 4]) + m[..., 5]
        off = m[..., 0] + np.maximum(m[..., 1], m[..., 2] + np.maximum(m[..., 3], m[..., 4] + m[..., 5]))
        return as_built, beside, off
    a, b, e = chains(inner + tris)
    c, d, ec = chains(inner4 + tris4)
    lit = tiles(inner + tris)[..., 1] > 0            # (a shadow ray was cast: some camera ray of the tile hit something)
    r = lambda v: "%6d (%.2f)" % (v.max(), v.max() / a.max())
    print("f%-4d | %9d | %6d | %s | %s | %s | %s | %s | %6.0f | %d / %d" % (f, int(lit.sum()), a.max(), r(b), r(c), r(d), r(e), r(ec), a[lit].mean(), sp2, sp4))
    if f == frames[0]:
        t = np.unravel_index(a.argmax(), a.shape)
        m = tiles(inner + tris)[t]
        print("      the longest tile (%d, %d): longest ray per generation camera / shadow / reflection / shadow / reflection / shadow = %s" % (t[1], t[0], m[:6].tolist()))
        share = tris[..., :6].sum() / max(1, (inner + tris)[..., :6].sum())
        print("      triangles are %.0f %% of all steps; the four-wide walk visits %.2f of the binary walk's inner records and tests %.2f of its triangles" % (
            100 * share, inner4[..., :6].sum() / max(1, inner[..., :6].sum()), tris4[..., :6].sum() / max(1, tris[..., :6].sum())))
