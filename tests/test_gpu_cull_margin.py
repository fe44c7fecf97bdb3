"""How close does an accepted hit ever come to the ordered walk's cull bound?  (k_raytrace.hip ordered walk)

The ordered walk skips a subtree when a LOWER bound of the ray's entry into the subtree's box, grown by a slack, exceeds
sqrt(best hit so far) * 1.001 + delta.  That is safe as long as the bound really is a lower bound of the distance of every hit
inside the box: near_g(box) <= t(hit) for every box on the path from the root to the hit triangle's leaf.  The fuzz tests
show frames never differ; this test measures the margin itself on the device: for tens of thousands of rays it takes the
oracle's closest hit, asks the device for the bounds of every box above the hit triangle exactly as a lane computes them
(mi355i_cull_probe runs ray_box_fast_ordered on the pairs), and asserts
  * near_g <= t for every pair -- no accepted hit could be culled even with NO slack in the distance bound, and
  * the part of the box slack (dmax) that the rounding of hit points and box tests actually consumes stays below a floor."""
import ctypes as C
import os

import numpy as np
import pytest

import renderer_amd as R

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def ancestors(nodes):
    """parent[] of the reference's pre-order node array and, per triangle-list position, the leaf that holds it"""
    n = nodes.shape[0]
    a, b = nodes[:, 6].astype(np.int64), nodes[:, 7].astype(np.int64)
    leaf = (a & 0x80000000) != 0
    parent = np.full(n, -1, np.int64)
    inner = np.flatnonzero(~leaf)
    parent[a[inner]] = inner
    parent[b[inner]] = inner
    cnt, first = a[leaf] & 0x7fffffff, b[leaf]
    leaf_of = np.full(int((first + cnt).max()), -1, np.int64)
    for node, f, c in zip(np.flatnonzero(leaf), first, cnt):
        leaf_of[f:f + c] = node
    return parent, leaf_of


def probe(scene, rays, pair_ray, pair_box):
    f = R.lib().mi355i_cull_probe
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    out = np.zeros((len(pair_ray), 4), np.float32)
    rc = f(scene.context(), rays.ctypes.data, len(rays), pair_ray.ctypes.data, pair_box.ctypes.data, len(pair_ray), out.ctypes.data)
    assert rc == 0, R.lib().mi355_last_error().decode()
    return out


@pytest.mark.parametrize("mesh", ["dragon_vis.ply", "chessboard.tri", "statue.ply"])
def test_no_accepted_hit_comes_near_the_cull_bound(mesh, oracle, oracle_scene):
    from oracle import refcore as RC                     # (only its numpy restatement of the primary rays is used)
    s = R.Scene(R.assets.mesh_path(mesh)); s.bvh_create()
    osc = oracle_scene(mesh, True)
    nodes, idx = s.bvh_arrays()
    boxes = nodes[:, :6].view(np.float32)
    parent, leaf_of = ancestors(nodes)
    pos_of_tri = np.empty(len(idx), np.int64); pos_of_tri[idx] = np.arange(len(idx))
    rng = np.random.default_rng(3)
    sets = []
    for frame in (0, 57, 140):                           # camera rays of three orbit positions
        cam = oracle.benchmark_frame(frame)[0]
        sets.append(RC.primary_rays(cam, 256, 144, 288))
    # rays between random points of the scene's box (origins inside the model, grazing directions, every octant)
    lo, hi = boxes[0, :3], boxes[0, 3:]
    o = rng.uniform(lo, hi, (20000, 3)); t = rng.uniform(lo, hi, (20000, 3))
    d = (t - o) / np.linalg.norm(t - o, axis=1, keepdims=True)
    sets.append(np.concatenate([o, d], axis=1).astype(np.float32))
    rays = np.ascontiguousarray(np.concatenate(sets).astype(np.float32))
    tri, hit = osc.trace_hits(rays)
    ok = np.flatnonzero(tri >= 0)
    assert len(ok) > 3000
    # distance of the hit as the kernel has it: sqrtf(distancesq(origin, hit)) in float32
    dv = (hit[ok] - rays[ok, :3]).astype(np.float32)
    t_hit = np.sqrt((dv[:, 0] * dv[:, 0] + dv[:, 1] * dv[:, 1] + dv[:, 2] * dv[:, 2]).astype(np.float32)).astype(np.float32)
    pr, pb, pt = [], [], []
    cur = leaf_of[pos_of_tri[tri[ok]]]
    live = np.arange(len(ok))
    while len(live):
        pr.append(ok[live]); pb.append(boxes[cur[live]]); pt.append(t_hit[live])
        cur[live] = parent[cur[live]]
        live = live[cur[live] >= 0]
    pair_ray = np.ascontiguousarray(np.concatenate(pr).astype(np.uint32))
    pair_box = np.ascontiguousarray(np.concatenate(pb).astype(np.float32))
    t_pair = np.concatenate(pt)
    out = probe(s, rays, pair_ray, pair_box)
    near_g, far_g, dmax, flags = out[:, 0], out[:, 1], out[:, 2], out[:, 3].astype(np.int32)
    tame = (flags & 4) != 0
    assert tame.mean() > 0.95                             # (untame rays take the exact test and are not culled by this bound)
    near_g, far_g, dmax, t_pair = near_g[tame], far_g[tame], dmax[tame], t_pair[tame]
    # 1. the bound is a bound: the hit lies at or beyond the entry bound and at or before the exit bound of every box above it
    worst = float((near_g - t_pair).max())
    assert worst <= 0.0, "an accepted hit lies %g BEFORE the lower bound of a box that contains it" % worst
    assert float((t_pair - far_g).max()) <= 0.0
    # 2. how much of the slack is ever used: entry of the un-grown box (near_g + dmax) beyond the hit distance, as a
    #    fraction of dmax -- 0 = the hit is inside its boxes as exact arithmetic says, 1 = the slack is exhausted
    used = np.maximum(near_g + dmax - t_pair, 0.0) / dmax
    hist = np.histogram(used, bins=[0, 1e-6, 1e-4, 1e-3, 1e-2, 0.05, 0.25, 1.0, np.inf])[0]
    print("%s: %d (hit, box) pairs; slack used: =0..1e-6: %d, ..1e-4: %d, ..1e-3: %d, ..1e-2: %d, ..0.05: %d, ..0.25: %d, ..1: %d, >1: %d; max %.4g"
          % ((mesh, len(used)) + tuple(hist) + (float(used.max()),)))
    assert float(used.max()) < 0.05, "the hit-point rounding consumes %.3g of the box slack (floor: 5 %%)" % float(used.max())
