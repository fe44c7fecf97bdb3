#!/usr/bin/env python3
"""Phase profile of the tiled rasterizer's k_rs_tile (counting frames): cycles between the barriers summed over the blocks,
work counts, and the frame rates of single frames."""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import renderer_amd as R
L = R.lib()
L.mi355i_fetch_profile.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
W, H = 1920, 1080
mesh = sys.argv[1] if len(sys.argv) > 1 else "chessboard.tri"
s = R.Scene(R.assets.mesh_path(mesh))
cam, lights, n = R.benchmark_frame(0)
s.shadowmap_render(0, lights[0])
names = ["bins+clear", "filter", "stage", "depth", "runs", "attr", "shade", "tile_total", "active_tiles", "longest_tile", "kept", "items", "runs", "entries_read"]
for mode in (4, 6, 8):
    img, _, st = s.render(mode, cam, lights, n, R.default_opts(W, H, collect_stats=1))
    prof = (C.c_uint64 * 20)()
    L.mi355i_fetch_profile(s.context(), prof)
    p = [int(v) for v in prof]
    nb = max(1, p[8])
    print("mode %d: kernel_ms %.3f (counting frame)" % (mode, st.kernel_ms))
    print("   per active block, cycles: " + ", ".join("%s %.0f" % (names[i], p[i] / nb) for i in range(8)))
    print("   longest tile: %d cycles, %d bin entries read, %d runs" % (p[9] >> 32, (p[9] >> 16) & 0xffff, p[9] & 0xffff))
    print("   tiles with entries %d, longest %d cycles; kept %d (%.1f / tile), depth items %d (%.1f), runs %d (%.1f), bin entries read %d (%.1f); background tiles %d, %.0f cycles each" % (
        p[8], p[9] >> 32, p[10], p[10] / nb, p[11], p[11] / nb, p[12], p[12] / nb, p[13], p[13] / nb, p[15], p[14] / max(1, p[15])))
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream(dev)
buf = torch.zeros((H, W), dtype=torch.int32, device=dev)
cams = [R.benchmark_frame(k) for k in range(200)]
out = {}
for nt in (64, 128, 256, 512):
  for mode in (4, 6, 8):
    o = R.default_opts(W, H)
    o.tune[3] = nt
    for k in range(5): s.render_device(mode, *cams[k], o, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
    torch.cuda.synchronize(dev); t = time.perf_counter()
    for k in range(200): s.render_device(mode, *cams[k], o, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
    torch.cuda.synchronize(dev); out["mode%d_nt%d_fps" % (mode, nt)] = round(200 / (time.perf_counter() - t), 1)
print(json.dumps(out))
