#!/usr/bin/env python3
"""The synchronous seam of the rasterizer (mi355_render into a page-locked canvas, one frame per call along the orbit) with and
without mi355_opts::keep_canvas: frames/s, and render_cli -b -p 1 with and without --keep-canvas.

    python scripts/keep_canvas_rate.py [frames]"""
import json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import renderer_amd as R

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
W, H = 1920, 1080
out = {}
for mesh in ("chessboard.tri", "dragon_vis.ply"):
    s = R.Scene(R.assets.mesh_path(mesh))
    cams = [R.benchmark_frame(k) for k in range(200)]
    s.shadowmap_render(0, cams[0][1][0])
    canvas = R.host_array((H, W))
    for mode in (6, 8):
        for keep in (0, 1):
            o = R.default_opts(W, H, keep_canvas=keep)
            for k in range(20): s.render_into(mode, *cams[k], o, canvas)
            best = 0.0
            for rep in range(3):
                t = time.perf_counter()
                for k in range(n): s.render_into(mode, *cams[k % 200], o, canvas)
                best = max(best, n / (time.perf_counter() - t))
            out["%s mode %d keep_canvas=%d" % (mesh.split(".")[0], mode, keep)] = round(best)
    R.host_array_free(canvas)
# three frames in flight (mi355_render_async / _wait), a canvas per slot
s = R.Scene(R.assets.mesh_path("chessboard.tri"))
cams = [R.benchmark_frame(k) for k in range(200)]
s.shadowmap_render(0, cams[0][1][0])
ring = [R.host_array((H, W)) for _ in range(3)]
for mode in (6, 8):
    for keep in (0, 1):
        o = R.default_opts(W, H, keep_canvas=keep)
        best = 0.0
        for rep in range(4):
            tickets = [None] * 3
            t = time.perf_counter()
            for k in range(n):
                if tickets[k % 3] is not None: s.render_wait(tickets[k % 3])
                tickets[k % 3] = s.render_async(mode, *cams[k % 200], o, ring[k % 3])
            for tk in tickets: s.render_wait(tk)
            if rep: best = max(best, n / (time.perf_counter() - t))
        out["chessboard mode %d three in flight keep_canvas=%d" % (mode, keep)] = round(best)
for c in ring: R.host_array_free(c)
# the raytracer (dragon, depth 3): the synchronous call and three in flight
s = R.Scene(R.assets.mesh_path("dragon_vis.ply")); s.bvh_create()
ring = [R.host_array((H, W)) for _ in range(3)]
nr = max(100, n // 4)
for keep in (0, 1):
    o = R.default_opts(W, H, keep_canvas=keep)
    for k in range(10): s.render_into(9, *cams[k], o, ring[0])
    best = 0.0
    for rep in range(3):
        t = time.perf_counter()
        for k in range(nr): s.render_into(9, *cams[k % 200], o, ring[0])
        best = max(best, nr / (time.perf_counter() - t))
    out["dragon mode 9 keep_canvas=%d" % keep] = round(best)
    best = 0.0
    for rep in range(4):
        tickets = [None] * 3
        t = time.perf_counter()
        for k in range(nr):
            if tickets[k % 3] is not None: s.render_wait(tickets[k % 3])
            tickets[k % 3] = s.render_async(9, *cams[k % 200], o, ring[k % 3])
        for tk in tickets: s.render_wait(tk)
        if rep: best = max(best, nr / (time.perf_counter() - t))
    out["dragon mode 9 three in flight keep_canvas=%d" % keep] = round(best)
for c in ring: R.host_array_free(c)
cli = os.path.join(os.path.dirname(R.RENDER_SO), "render_cli")
for flag in (["-p", "1"], ["-p", "1", "--keep-canvas"], ["-p", "3"], ["-p", "3", "--keep-canvas"]):
    r = subprocess.run([cli, "-b", "-n", "2000", "-m", "6", "-W", "1920", "-H", "1080"] + flag + [R.assets.mesh_path("chessboard.tri")], capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("Rendering")]
    out["render_cli -b -m 6 " + " ".join(flag)] = line[-1] if line else r.stderr[-300:]
for args in (["-m", "9", "-W", "1920", "-H", "1080"], ["-m", "9", "-W", "3840", "-H", "2160"]):
    for flag in (["-p", "1"], ["-p", "1", "--keep-canvas"], ["-p", "3"], ["-p", "3", "--keep-canvas"]):
        r = subprocess.run([cli, "-b", "-n", "400"] + args + flag + [R.assets.mesh_path("dragon_vis.ply")], capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("Rendering")]
        out["render_cli -b dragon " + " ".join(args + flag)] = line[-1] if line else r.stderr[-300:]
print(json.dumps(out, indent=1))
