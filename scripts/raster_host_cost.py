import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import renderer_amd as R
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream(dev)
W, H = 1920, 1080
s = R.Scene(R.assets.mesh_path("chessboard.tri"))
cams = [R.benchmark_frame(k) for k in range(200)]
buf = torch.zeros((H, W), dtype=torch.int32, device=dev)
for label, o in (("pipelined", R.default_opts(W, H)), ("one stream", R.default_opts(W, H, tune=R.tune(nopipe=1)))):
    for k in range(20): s.render_device(6, *cams[k], o, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(200): s.render_device(6, *cams[k], o, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
    t1 = time.perf_counter(); torch.cuda.synchronize(dev); t2 = time.perf_counter()
    print("%s: enqueue %.1f us per frame, total %.1f us per frame" % (label, (t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
