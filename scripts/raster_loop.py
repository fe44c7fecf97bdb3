#!/usr/bin/env python3
"""N single frames of one raster mode (chessboard 1080p) on the device path: the workload behind the rocprofv3 counter passes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import renderer_amd as R
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1
W, H = (int(os.environ.get("RL_W", "1920")), int(os.environ.get("RL_H", "1080")))
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream(dev)
s = R.Scene(R.assets.mesh_path("chessboard.tri"))
cams = [R.benchmark_frame(k) for k in range(200)]
s.shadowmap_render(0, cams[0][1][0])
bufs = [torch.zeros((H, W), dtype=torch.int32, device=dev) for _ in range(batch)]
o = R.default_opts(W, H)
t = time.perf_counter()
for i in range(n):
    if batch == 1:
        s.render_device(mode, *cams[i % 200], o, bufs[0].data_ptr(), W * 4, 0, stream.cuda_stream)
    else:
        fs = [(batch * i + j) % 200 for j in range(batch)]
        s.render_batch_device(mode, [cams[f][0] for f in fs], [cams[f][1] for f in fs], 1, o, [b.data_ptr() for b in bufs], W * 4, None, stream.cuda_stream)
torch.cuda.synchronize(dev)
print("mode %d: %d x %d frames, %.1f fps" % (mode, n, batch, n * batch / (time.perf_counter() - t)))
