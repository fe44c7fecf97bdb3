import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import renderer_amd as R
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream(dev)
W, H = 640, 360
s = R.Scene(R.assets.mesh_path("chessboard.tri"))
cam, lights, n = R.benchmark_frame(0)
buf = torch.zeros((H, W), dtype=torch.int32, device=dev)
for name, tn in (("one_stream", R.tune(nopipe=1)), ("overlapped", R.tune())):
    o = R.default_opts(W, H, tune=tn)
    for k in range(3):
        t = time.perf_counter()
        s.render_device(6, cam, lights, n, o, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
        torch.cuda.synchronize(dev)
        print(name, k, "%.3f ms" % ((time.perf_counter() - t) * 1e3), int((buf != 0).sum()), flush=True)
    try:
        s.fetch_stats(); print("stats ok")
    except Exception as e:
        print("stats:", e)
img, _, st = s.render(6, cam, lights, n, R.default_opts(W, H))
print("sync render", int((img != 0).sum()), st.kernel_ms)
