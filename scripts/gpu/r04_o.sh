mkdir -p gpurun_out
(timeout 600 python -m pytest tests -m gpu -q -x --capture=sys -k "async or batch or tile_cull or frame_overlap or parity or render_cli or threads or lifecycle" 2>&1 | tail -4) > gpurun_out/r04o_pytest.log; tail -3 gpurun_out/r04o_pytest.log
cat > /tmp/seam.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, renderer_amd as R
W, H = 1920, 1080
s = R.Scene(R.assets.mesh_path("dragon_vis.ply")); s.bvh_create()
cams = [R.benchmark_frame(k) for k in range(200)]
o = R.default_opts(W, H)
a, b = np.zeros((H, W), np.uint32), np.zeros((H, W), np.uint32)
s.host_register(b)
for name, buf in (("pageable", a), ("registered", b)):
    for k in range(5): s.render_into(9, *cams[k], o, buf)
    t = time.perf_counter(); kms = 0
    for k in range(200): kms += s.render_into(9, *cams[k], o, buf).kernel_ms
    print(name, "%.1f fps, kernel %.4f ms" % (200 / (time.perf_counter() - t), kms / 200), "nonblack", int((buf != 0).sum()))
print("same", bool(np.array_equal(a, b)))
PY
timeout 120 python /tmp/seam.py 2>&1 | grep -v amdgpu > gpurun_out/r04o_seam.log; cat gpurun_out/r04o_seam.log
MI355_NO_ZERO_COPY=1 timeout 120 python /tmp/seam.py 2>&1 | grep -v amdgpu | head -2
timeout 300 bash scripts/render_cli_configs.sh > gpurun_out/r04o_cli.log 2>&1; cat gpurun_out/r04o_cli.log
