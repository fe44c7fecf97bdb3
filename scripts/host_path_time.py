#!/usr/bin/env python3
"""Frames per second through the host-buffer entry points (dragon 1080p raytrace, chessboard 1080p Phong):
mi355_render into pageable memory, into a registered buffer, and pipelined (mi355_render_async, 3 in flight)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd as R


def measure(mesh, mode, n=120):
    W, H = 1920, 1080
    s = R.Scene(R.assets.mesh_path(mesh))
    if mode >= 9:
        s.bvh_update()
    o = R.default_opts(W, H)
    cams = [R.benchmark_frame(k) for k in range(n)]
    out = {}
    buf = np.zeros((H, W), np.uint32)
    for c in cams[:5]: s.render_into(mode, *c, o, buf)
    t0 = time.perf_counter(); ks = 0.0
    for c in cams: ks += s.render_into(mode, *c, o, buf).kernel_ms
    out["sync_pageable_fps"] = round(n / (time.perf_counter() - t0), 1); out["kernel_ms"] = round(ks / n, 4)
    bufs = [np.zeros((H, W), np.uint32) for _ in range(4)]
    for b in bufs: s.host_register(b)
    for c in cams[:5]: s.render_into(mode, *c, o, bufs[0])
    t0 = time.perf_counter()
    for c in cams: s.render_into(mode, *c, o, bufs[0])
    out["sync_registered_fps"] = round(n / (time.perf_counter() - t0), 1)
    for depth in (2, 3, 4):
        t0 = time.perf_counter(); q = []
        for k, c in enumerate(cams):
            if len(q) == depth: s.render_wait(q.pop(0))
            q.append(s.render_async(mode, *c, o, bufs[k % 4]))
        for t in q: s.render_wait(t)
        out["pipelined_%d_registered_fps" % depth] = round(n / (time.perf_counter() - t0), 1)
    for b in bufs: s.host_unregister(b)
    pb = [np.zeros((H, W), np.uint32) for _ in range(3)]
    t0 = time.perf_counter(); q = []
    for k, c in enumerate(cams):
        if len(q) == 3: s.render_wait(q.pop(0))
        q.append(s.render_async(mode, *c, o, pb[k % 3]))
    for t in q: s.render_wait(t)
    out["pipelined_3_pageable_fps"] = round(n / (time.perf_counter() - t0), 1)
    return out


if __name__ == "__main__":
    print(json.dumps({"dragon_raytrace_1080p": measure("dragon_vis.ply", 9), "chessboard_phong_1080p": measure("chessboard.tri", 6)}))
