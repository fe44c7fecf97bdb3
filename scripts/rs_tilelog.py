#!/usr/bin/env python3
"""When the blocks of a raster frame's three kernels start, pass their phases and end -- in the production schedule (three frames in
flight) and for a frame by itself.  Needs the measuring variant:

    bash scripts/build_rs_variant.sh tilelog -DRS_TILELOG=1
    MI355_WAVELOG=1 MI355_RENDER_SO=renderer_amd/lib/variant_tilelog.so python scripts/rs_tilelog.py [mode]

The library keeps the logs of the last four overlapped frames side by side (2 048 block records of 16 words each; 100 MHz clock)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import renderer_amd as R
dev = torch.device("cuda", 0)
W, H = 1920, 1080
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 6
s = R.Scene(R.assets.mesh_path("chessboard.tri"))
cams = [R.benchmark_frame(k) for k in range(200)]
s.shadowmap_render(0, cams[0][1][0])
st = torch.cuda.current_stream(dev)
buf = torch.zeros((H, W), dtype=torch.int32, device=dev)
f = R.lib().mi355i_fetch_wave_profiles
f.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
o = R.default_opts(W, H)
q = lambda v: [round(float(np.percentile(v, p)), 1) for p in (0, 10, 50, 90, 99, 100)] if len(v) else []


def fetch():
    out = (C.c_ulonglong * (16 * 8192))()
    n = f(s.context(), out, 8192)
    return np.array(out[: 16 * n], dtype=np.uint64).reshape(n, 16).astype(np.float64)


def report(tag, a, t_ref=None):
    """a: the 2 048 records of one frame: [0] start, [1..8] time of the first tile's phases, [9] end, [10] entries | kept << 32, [11] tile"""
    tile = a[a[:, 9] > 0]
    busy = tile[tile[:, 2] > 0]                                   # blocks that drew a tile
    setup = a[a[:, 13] > 0][:, 12:14]; fill = a[a[:, 15] > 0][:, 14:16]
    t0 = setup[:, 0].min() if len(setup) else tile[:, 0].min()
    us = lambda v: (v - t0) / 100.0
    names = ["taken", "cleared", "filtered", "staged", "depth", "runs", "attr", "shaded"]
    life = busy[:, 1:9].sum(axis=1) / 100.0
    d = {"frame": tag, "frame_start_us_abs": round(float((t0 - (t_ref or t0)) / 100.0), 1),
         "setup_end_us": q(us(setup[:, 1])), "fill_start_us": q(us(fill[:, 0])), "fill_end_us": q(us(fill[:, 1])),
         "busy_tiles": int(len(busy)), "tile_start_us": q(us(tile[:, 0])), "busy_tile_end_us": q(us(busy[:, 0]) + life),
         "kernel_end_us": round(float(us(tile[:, 9]).max()), 1), "busy_tile_life_us": q(life), "sum_busy_life_us": round(float(life.sum()))}
    fa = a[:, 14:16]; ok = fa[:, 1] > 0
    if ok.any():
        fe = np.where(ok, (fa[:, 1] - t0) / 100.0, 0.0)
        slow = np.argsort(-fe)[:6]
        d["slowest_fill_blocks"] = [[int(i), round(float((fa[i, 0] - t0) / 100.0), 1), round(float(fe[i]), 1)] for i in slow]     # index, start, end
        sa = a[:, 12:14]; oks = sa[:, 1] > 0
        se = np.where(oks, (sa[:, 1] - t0) / 100.0, 0.0)
        d["slowest_setup_blocks"] = [[int(i), round(float((sa[i, 0] - t0) / 100.0), 1), round(float(se[i]), 1)] for i in np.argsort(-se)[:4]]
    for i, nm in enumerate(names):
        d["phase_" + nm + "_us"] = q(busy[:, 1 + i] / 100.0)
    d["phase_means_us"] = {nm: round(float(busy[:, 1 + i].mean() / 100.0), 2) for i, nm in enumerate(names)}
    d["entries_read_mean_max"] = [round(float((busy[:, 10] % 2 ** 32).mean()), 1), float((busy[:, 10] % 2 ** 32).max())]
    order = np.argsort(-life)[:3]
    d["longest"] = [{"life_us": round(float(life[i]), 1), "entries": int(busy[i, 10] % 2 ** 32), "kept": int(busy[i, 10] // 2 ** 32),
                     "phases_us": [round(float(busy[i, 1 + k] / 100.0), 1) for k in range(8)]} for i in order]
    print(json.dumps(d), flush=True)
    return t0


# a frame by itself (synchronous calls take the caller's stream; the device entry point with a sync in between is a lone overlapped frame)
for k in (0, 100):
    for rep in range(3):
        s.render_device(mode, *cams[k], o, buf.data_ptr(), W * 4, 0, st.cuda_stream); torch.cuda.synchronize(dev)
    a = fetch()
    last = max(range(4), key=lambda g: a[g * 2048:(g + 1) * 2048, 9].max())
    report("alone f%d" % k, a[last * 2048:(last + 1) * 2048])
# the production schedule: 400 frames back to back, the last four logged
for k in range(400): s.render_device(mode, *cams[k % 200], o, buf.data_ptr(), W * 4, 0, st.cuda_stream)
torch.cuda.synchronize(dev)
a = fetch()
groups = sorted(range(4), key=lambda g: a[g * 2048:(g + 1) * 2048, 0][a[g * 2048:(g + 1) * 2048, 0] > 0].min())
t_ref = None
for g in groups:
    t0 = report("in flight (group %d)" % g, a[g * 2048:(g + 1) * 2048], t_ref)
    t_ref = t_ref or t0
