import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import renderer_amd as R
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream(dev)
W, H = 1920, 1080
s = R.Scene(R.assets.mesh_path("chessboard.tri"))
cams = [R.benchmark_frame(k) for k in range(200)]
s.shadowmap_render(0, cams[0][1][0])
buf = torch.zeros((H, W), dtype=torch.int32, device=dev)
for nt in (128, 192, 256, 320, 384, 512):
    t = [0] * 8; t[3] = nt
    o = R.default_opts(W, H, tune=tuple(t))
    for k in range(20): s.render_device(6, *cams[k], o, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
    torch.cuda.synchronize(dev); t0 = time.perf_counter()
    for k in range(200): s.render_device(6, *cams[k], o, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
    torch.cuda.synchronize(dev); print("nt %d: %.0f fps" % (nt, 200 / (time.perf_counter() - t0)))
