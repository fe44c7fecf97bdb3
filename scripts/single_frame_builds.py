import sys, os, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import renderer_amd as R
s = R.Scene(R.assets.mesh_path("dragon_vis.ply")); s.bvh_create()
for (W, H, mode) in ((3840, 2160, 9), (1920, 1080, 10), (1920, 1080, 9), (640, 360, 9)):
    for bpc in (0, 2, 3, 4):
        o = R.default_opts(W, H, tune=R.tune(bpc=bpc))
        ms = []
        for k in list(range(0, 200, 20)) * 2:
            cam, lights, n = R.benchmark_frame(k)
            ms.append(s.render(mode, cam, lights, n, o)[2].kernel_ms)
        print(json.dumps({"W": W, "H": H, "mode": mode, "bpc": bpc, "kernel_ms": round(float(np.mean(ms[10:])), 4)}), flush=True)
