#!/usr/bin/env python3
"""Randomised self-consistency sweep (not part of the test suite): for random meshes, cameras, modes and frame sizes, (a) a
frame rendered as interleaved bands with random band heights / counts, compact or in place, must reassemble to the unsharded
frame, and (b) a batch of frames (raytrace and raster modes, with and without bands) must equal the frames rendered one by one."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import renderer_amd as R
from renderer_amd import multigpu

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=40)
ap.add_argument("--seed", type=int, default=1)
args = ap.parse_args()
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev)
scenes = {}
for name in ("dragon_vis.ply", "chessboard.tri", "legocar.3ds"):
    s = R.Scene(R.assets.mesh_path(name)); s.bvh_create("device"); scenes[name] = s
bad = 0
for it in range(args.n):
    rng = np.random.default_rng(args.seed * 15485863 + it)
    name = list(scenes)[int(rng.integers(0, 3))]
    s = scenes[name]
    mode = int(rng.choice([1, 2, 4, 5, 6, 7, 8, 9, 9, 10]))
    W, H = [(320, 240), (333, 217), (97, 64), (640, 41), (8, 8)][int(rng.integers(0, 5))]
    frames = [int(x) for x in rng.integers(0, 200, int(rng.integers(2, 6)))]
    cl = [R.benchmark_frame(f, bool(rng.integers(0, 2))) for f in frames]
    nl = cl[0][2]
    cl = [c for c in cl if c[2] == nl] or cl[:1]
    if mode in (7, 8):
        s.shadowmap_render(0, cl[0][1][0])
        cl = [c for c in cl if c[2] == 1][:1] or [R.benchmark_frame(frames[0])]   # one light position per shadow map
        nl = 1
    # (a) bands of frame 0
    cam, lights, _ = cl[0]
    full = s.render(mode, cam, lights, nl, R.default_opts(W, H))[0]
    br, bc, compact = int(rng.integers(1, 41)), int(rng.integers(1, 6)), int(rng.integers(0, 2))
    parts = []
    for r in range(bc):
        o = R.default_opts(W, H, band_rows=br, band_index=r, band_count=bc, compact_rows=compact)
        parts.append(s.render(mode, cam, lights, nl, o)[0])
    if compact:
        got = multigpu.assemble_numpy(parts, H, br) if bc > 1 else parts[0]
    else:
        got = np.zeros_like(full)
        ys = np.arange(H)
        for r in range(bc):
            mine = (ys // br) % bc == r
            got[mine] = parts[r][mine]
            if parts[r][~mine].any(): bad += 1; print("case %d: rows of other bands were written (mode %d)" % (it, mode), flush=True)
    if got.shape != full.shape or (got != full).any():
        bad += 1; print("case %d %s mode %d %dx%d bands %d x %d rows compact %d: differs" % (it, name, mode, W, H, bc, br, compact), flush=True)
    # (b) batch vs singles
    if mode >= 4:
        bufs = [torch.zeros((H, W), dtype=torch.int32, device=dev) for _ in cl]
        try:
            s.render_batch_device(mode, [c[0] for c in cl], [c[1] for c in cl], nl, R.default_opts(W, H), [b.data_ptr() for b in bufs], W * 4, None, stream.cuda_stream)
            torch.cuda.synchronize(dev)
            s.fetch_stats()
            for j, c in enumerate(cl):
                one = s.render(mode, c[0], c[1], nl, R.default_opts(W, H))[0]
                if (bufs[j].cpu().numpy().astype(np.uint32) != one).any():
                    bad += 1; print("case %d %s mode %d %dx%d batch of %d: frame %d differs" % (it, name, mode, W, H, len(cl), j), flush=True)
        except R.Mi355Error as e:
            bad += 1; print("case %d %s mode %d batch error: %s" % (it, name, mode, e), flush=True)
print("bands/batches fuzz: %d cases, %d bad" % (args.n, bad))
