#!/usr/bin/env python3
"""When the waves of a batched launch start, run dry and end (variant built with -DRT_WAVELOG=1, run with MI355_WAVELOG=1 and
MI355_RENDER_SO=.../variant_wavelog.so): the launch runs by itself on the caller's stream; times in us from the first wave's start."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import renderer_amd as R
dev = torch.device("cuda", 0)
W, H, B = 1920, 1080, int(os.environ.get("RT_B", "8"))
s = R.Scene(R.assets.mesh_path("dragon_vis.ply")); s.bvh_create()
o = R.default_opts(W, H, max_ray_depth=3, tune=R.tune(**json.loads(os.environ.get("RT_TUNE", "{}"))))
bufs = [torch.zeros((H, W), dtype=torch.int32, device=dev) for _ in range(B)]
st = torch.cuda.current_stream(dev)
cams = [R.benchmark_frame(k) for k in range(200)]
f = R.lib().mi355i_fetch_wave_profiles
f.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
for first in (0, 40, 96):
    ks = [(first + j) % 200 for j in range(B)]
    for rep in range(2):
        s.render_batch_device(9, [cams[k][0] for k in ks], [cams[k][1] for k in ks], 1, o, [b.data_ptr() for b in bufs], W * 4, None, st.cuda_stream)
        torch.cuda.synchronize(dev)
    out = (C.c_ulonglong * (16 * 8192))()
    n = f(s.context(), out, 8192)
    a = np.array(out[: 16 * n], dtype=np.uint64).reshape(n, 16).astype(np.float64)
    a = a[a[:, 2] > 0]
    t0 = a[:, 0].min()
    start, dry, end = (a[:, 0] - t0) / 100.0, (np.where(a[:, 1] > 0, a[:, 1], a[:, 2]) - t0) / 100.0, (a[:, 2] - t0) / 100.0
    q = lambda v: [round(float(np.percentile(v, p)), 1) for p in (0, 10, 50, 90, 99, 100)]
    print(json.dumps({"frames": ks[0], "waves": int(len(a)), "start_us_p0_10_50_90_99_100": q(start), "dry_us": q(dry), "end_us": q(end),
                      "mean_life_over_span": round(float((end - start).mean() / end.max()), 3),
                      "mean_busy_until_dry_over_span": round(float((dry - start).mean() / end.max()), 3)}), flush=True)
