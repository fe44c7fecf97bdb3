#!/usr/bin/env python3
"""Is the frame-by-frame raster rate the HOST's or the GPU's?  The same calls (chessboard 1080p, mode 6, device entry point) with
the camera turned AWAY from the mesh: every triangle is rejected by k_rs_setup, no tile holds anything, the kernels are as short as
kernels get -- what remains is the host's rate for a frame's calls.  Then the benchmark orbit for comparison."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import renderer_amd as R
W, H, n = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 8000
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream(dev)
s = R.Scene(R.assets.mesh_path("chessboard.tri"))
buf = torch.zeros((H, W), dtype=torch.int32, device=dev)
o = R.default_opts(W, H)
cams = [R.benchmark_frame(k) for k in range(200)]
away = R.camera([4.8, 0.0, 0.0], [9.6, 0.0, 0.0])
la = (R.Light * 2)(); la[0] = R.light([3.4, 3.4, 4.8], away)
for name, pick in (("camera turned away (host rate)", lambda i: (away, la, 1)), ("benchmark orbit", lambda i: cams[i % 200])):
    for rep in range(2):
        torch.cuda.synchronize(dev)
        t = time.perf_counter()
        for i in range(n):
            s.render_device(6, *pick(i), o, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
        t_enq = time.perf_counter() - t
        torch.cuda.synchronize(dev)
        t_all = time.perf_counter() - t
    print("%-32s %8.1f fps (%.1f us per frame; the calls alone returned after %.1f us per frame)" % (name, n / t_all, t_all / n * 1e6, t_enq / n * 1e6))
