#!/usr/bin/env python3
"""Scale check (not part of the test suite): a generated closed surface with 2 n^2 triangles -- GPU BVH builder vs host
builder byte for byte, then frames of several modes vs the oracle (which loads the .bvh cache the host layer wrote)."""
import argparse, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import renderer_amd as R
from oracle import oracle_ctypes as O

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=500, help="grid size: 2 n^2 triangles")
ap.add_argument("--modes", default="9,6,8,2")
args = ap.parse_args()
n = args.n
u, v = np.meshgrid(np.linspace(0, 2 * np.pi, n, endpoint=False), np.linspace(0, 2 * np.pi, n, endpoint=False), indexing="ij")
r = 0.35 + 0.05 * np.sin(7 * u) * np.cos(5 * v)
P = np.stack([(1 + r * np.cos(v)) * np.cos(u), (1 + r * np.cos(v)) * np.sin(u), r * np.sin(v) + 0.1 * np.sin(3 * u)], -1).reshape(-1, 3).astype(np.float32)
idx = np.arange(n * n).reshape(n, n)
a, b, c = idx, np.roll(idx, -1, 0), np.roll(idx, -1, 1)
d = np.roll(b, -1, 1)
F = np.concatenate([np.stack([a, b, c], -1).reshape(-1, 3), np.stack([b, d, c], -1).reshape(-1, 3)])
path = os.path.join(tempfile.mkdtemp(), "big%d.ply" % n)
with open(path, "w") as f:
    f.write("ply\nformat ascii 1.0\nelement vertex %d\nelement face %d\nend_header\n" % (len(P), len(F)))
    np.savetxt(f, np.concatenate([P, np.full((len(P), 1), 150)], 1), fmt="%.7g %.7g %.7g %d")
    np.savetxt(f, np.concatenate([np.full((len(F), 1), 3), F], 1), fmt="%d")
print("mesh: %d vertices, %d triangles" % (len(P), len(F)), flush=True)
O.build()
t = time.time(); h = R.Scene(path); h.bvh_create("host"); th = time.time() - t
t = time.time(); d = R.Scene(path); d.bvh_create("device"); td = time.time() - t
hn, hi = h.bvh_arrays(); dn, di = d.bvh_arrays()
same = hn.shape == dn.shape and bool((hn == dn).all()) and bool((hi == di).all())
print("BVH: %d nodes, depth %d; host builder %.2f s, device builder (incl. load) %.2f s; same tree: %s; ordered walk usable: %d"
      % (hn.shape[0], h.bvh_info()[2], th, td, same, d.walk_info()[0]), flush=True)
w = R.Scene(path); w.bvh_update(path)            # writes <mesh>.bvh, the cache in the reference's format
o = O.Scene(path); o.bvh_ensure(path + ".bvh")
bad = 0 if same else 1
for mode in [int(m) for m in args.modes.split(",")]:
    for frame in (0, 77):
        cam, lights, nl = R.benchmark_frame(frame); ocam, ol, on = O.benchmark_frame(frame)
        W, H = 640, 360
        maps = None
        if mode in (7, 8):
            maps = [o.shadowmap(ol[i]) for i in range(on)]
            for i in range(nl): d.shadowmap_render(i, lights[i])
        img, f32, st = d.render(mode, cam, lights, nl, R.default_opts(W, H), want_f32=mode >= 9)
        oi, of32, ost = o.render(mode, ocam, ol, on, O.default_opts(W, H, threads=os.cpu_count() or 1), shadow_maps=maps, want_f32=mode >= 9)
        diff = int((img != oi).sum()) + (int((f32 != of32).sum()) if mode >= 9 else 0)
        print("mode %d frame %d: %d differences, kernel %.3f ms" % (mode, frame, diff, st.kernel_ms), flush=True)
        bad += diff != 0
print("big scene check:", "OK" if bad == 0 else "%d FAILED" % bad)
