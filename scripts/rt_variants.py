#!/usr/bin/env python3
"""Round-3 raytrace variants on the bench workload (dragon 1080p, mode 9): batches of 8 frames per launch (frames/s, overlapped
launches as bench.py times them) and single frames (mi355_stats::kernel_ms of the synchronous call), for the work sharing inside a
wave (off / thresholds), the four-wide walk and the register builds.  One JSON line per variant; `steals` = subtrees handed from
lane to lane in the last single frame."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import renderer_amd as R
dev = torch.device("cuda", 0)
mesh = sys.argv[1] if len(sys.argv) > 1 else "dragon_vis.ply"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 3
W, H = 1920, 1080
s = R.Scene(R.assets.mesh_path(mesh)); s.bvh_create()
stream = torch.cuda.current_stream(dev)
cams = [R.benchmark_frame(k) for k in range(200)]
B = 8
bufs = [torch.zeros((H, W), dtype=torch.int32, device=dev) for _ in range(B)]
variants = [("default", {}), ("noshare", dict(noshare=1)), ("sharemin1", dict(sharemin=1)), ("sharemin4", dict(sharemin=4)), ("sharemin12", dict(sharemin=12)), ("sharemin16", dict(sharemin=16)), ("sharemin24", dict(sharemin=24)),
            ("sharemin32", dict(sharemin=32)), ("bpc3", dict(bpc=3)), ("bpc3 noshare", dict(bpc=3, noshare=1)), ("bpc3 sharemin16", dict(bpc=3, sharemin=16)), ("bpc3 sharemin4", dict(bpc=3, sharemin=4)), ("bpc2", dict(bpc=2)), ("bpc4", dict(bpc=4)),
            ("quad", dict(quad=1)), ("quad bpc3", dict(quad=1, bpc=3)), ("quad bpc2", dict(quad=1, bpc=2)),
            # bounds, not variants (other pictures): what is left when shadow rays / reflected rays cost nothing
            ("noshadows", dict(_opts=dict(use_shadows=0))), ("noshadows noshare", dict(noshare=1, _opts=dict(use_shadows=0))), ("norefl", dict(_opts=dict(use_reflections=0)))]
sel = os.environ.get("RT_VARIANTS")
if sel: variants = [v for v in variants if v[0] in sel.split(",")]
def prof12():
    out = (C.c_ulonglong * 20)()
    R.lib().mi355i_fetch_profile.argtypes = [C.c_void_p, C.c_void_p]
    return (int(out[12]), int(out[11])) if R.lib().mi355i_fetch_profile(s.context(), out) == 0 else (-1, -1)
ref_px = {}
T0 = time.perf_counter()
for label, t in variants:
    t = dict(t); extra = t.pop("_opts", {})
    o = R.default_opts(W, H, max_ray_depth=depth, tune=R.tune(**t), **extra)
    def step(i):
        ks = [(i * B + j) % 200 for j in range(B)]
        s.render_batch_device(9, [cams[k][0] for k in ks], [cams[k][1] for k in ks], 1, o, [b.data_ptr() for b in bufs], W * 4, None, stream.cuda_stream)
    for i in range(6): step(i)
    torch.cuda.synchronize(dev)
    best = 0.0
    for rep in range(3):
        t0 = time.perf_counter()
        for i in range(60): step(i)
        torch.cuda.synchronize(dev)
        best = max(best, 60 * B / (time.perf_counter() - t0))
    ms = []
    for k in list(range(0, 200, 10)) * 2:
        cam, lights, n = cams[k]
        _, _, st = s.render(9, cam, lights, n, o)
        ms.append(st.kernel_ms)
    ms = np.array(ms[20:])
    px, _, _ = s.render(9, *cams[0][:2], cams[0][2], o)
    key = json.dumps(extra, sort_keys=True)
    same = bool(np.array_equal(ref_px.setdefault(key, np.array(px)), np.array(px)))
    print(json.dumps({"variant": label, "t": round(time.perf_counter() - T0, 1), "same_pixels": same, "batch8_fps": round(best, 1), "single_ms_mean": round(float(ms.mean()), 4), "single_ms_min": round(float(ms.min()), 4),
                      "steals_events_last_frame": prof12()}), flush=True)
