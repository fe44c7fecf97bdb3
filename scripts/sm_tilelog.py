#!/usr/bin/env python3
"""Where the blocks of the shadow map's tile kernel spend their time (measuring variant: bash scripts/build_rs_variant.sh tilelog
-DRS_TILELOG=1; MI355_WAVELOG=1 MI355_RENDER_SO=renderer_amd/lib/variant_tilelog.so python scripts/sm_tilelog.py)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import renderer_amd as R
dev = torch.device("cuda", 0); st = torch.cuda.current_stream(dev)
f = R.lib().mi355i_fetch_wave_profiles
f.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
q = lambda v: [round(float(np.percentile(v, p)), 1) for p in (0, 10, 50, 90, 99, 100)]
for mesh in ("chessboard.tri", "dragon_vis.ply", "statue.ply"):
    s = R.Scene(R.assets.mesh_path(mesh))
    for k in range(3):
        s.light_update(0, [3.394, 3.394, 4.8], 1024, st.cuda_stream); torch.cuda.synchronize(dev)
    out = (C.c_ulonglong * (16 * 8192))()
    n = f(s.context(), out, 8192)
    a = np.array(out[: 16 * n], dtype=np.uint64).reshape(n, 16).astype(np.float64)[:1024]
    t0 = a[:, 0].min()
    names = ["clear", "scan", "collect", "draw", "store"]
    life = a[:, 1:6].sum(axis=1) / 100.0
    d = {"mesh": mesh, "start_us": q((a[:, 0] - t0) / 100.0), "life_us": q(life), "end_us": q((a[:, 0] - t0) / 100.0 + life),
         "means_us": {nm: round(float(a[:, 1 + i].mean() / 100.0), 2) for i, nm in enumerate(names)},
         "p99_us": {nm: round(float(np.percentile(a[:, 1 + i], 99) / 100.0), 2) for i, nm in enumerate(names)},
         "entries_seen": q(a[:, 6]), "kept": q(a[:, 7])}
    w = np.argsort(-life)[:3]
    d["longest"] = [{"tile": int(i), "life": round(float(life[i]), 1), "phases": [round(float(a[i, 1 + k] / 100.0), 1) for k in range(5)], "seen": int(a[i, 6]), "kept": int(a[i, 7])} for i in w]
    print(json.dumps(d), flush=True)
