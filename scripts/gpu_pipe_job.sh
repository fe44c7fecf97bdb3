timeout 150 python -m pytest tests/test_gpu_wire.py -x -q 2>&1 | tail -3
timeout 60 python - <<'PY' 2>&1 | grep -v amdgpu
import time, torch, renderer_amd as R
W, H = 1920, 1080
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream(dev)
s = R.Scene(R.assets.mesh_path("chessboard.tri"))
buf = torch.zeros((H, W), dtype=torch.int32, device=dev)
cams = [R.benchmark_frame(k) for k in range(100)]
o = R.default_opts(W, H)
for k in range(5): s.render_device(3, *cams[k], o, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
torch.cuda.synchronize(); t = time.perf_counter()
for k in range(100): s.render_device(3, *cams[k], o, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
torch.cuda.synchronize(); print("wireframe 1080p chessboard: %.1f fps" % (100 / (time.perf_counter() - t)))
PY
