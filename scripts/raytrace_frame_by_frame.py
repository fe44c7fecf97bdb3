#!/usr/bin/env python3
"""Raytraced frames one call at a time on one stream (dragon 1080p orbit, mi355_render_device): overlapped inside the library
(default) against one frame after the other (tune flag 32)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import renderer_amd as R
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
s = R.Scene(R.assets.mesh_path("dragon_vis.ply")); s.bvh_create()
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream(dev)
buf = torch.zeros((H, W), dtype=torch.int32, device=dev)
cams = [R.benchmark_frame(k) for k in range(200)]
for mode in (9, 10):
    for label, t in (("overlapped", {}), ("overlapped bpc3", dict(bpc=3)), ("overlapped bpc4", dict(bpc=4)), ("one_stream", dict(nopipe=1))):
        o = R.default_opts(W, H, tune=R.tune(**t))
        best = 0.0
        for rep in range(3):
            for k in range(5): s.render_device(mode, *cams[k], o, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
            torch.cuda.synchronize(dev); t0 = time.perf_counter()
            for k in range(200): s.render_device(mode, *cams[k], o, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
            t1 = time.perf_counter(); torch.cuda.synchronize(dev)
            best = max(best, 200 / (time.perf_counter() - t0))
        st = s.fetch_stats()
        print("mode %d %-16s %.1f fps (host enqueue %.1f us per frame; last frame %d + %d rays)" % (mode, label, best, (t1 - t0) / 200 * 1e6, st.normal_rays, st.shadow_rays))
