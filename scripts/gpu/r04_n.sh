mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q -x --capture=sys 2>&1 | tail -6) > gpurun_out/r04n_pytest.log; tail -4 gpurun_out/r04n_pytest.log
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
timeout 120 python /tmp/seam.py 2>&1 | grep -v amdgpu > gpurun_out/r04n_seam.log; cat gpurun_out/r04n_seam.log
MI355_FILL_IN_SELECT=1 timeout 120 python /tmp/seam.py 2>&1 | grep -v amdgpu > gpurun_out/r04n_seam_oldfill.log; cat gpurun_out/r04n_seam_oldfill.log
RT_VARIANTS=default timeout 200 python scripts/rt_variants.py 2>&1 | grep variant > gpurun_out/r04n_rt.log; cat gpurun_out/r04n_rt.log
MI355_FILL_IN_SELECT=1 RT_VARIANTS=default timeout 200 python scripts/rt_variants.py 2>&1 | grep variant >> gpurun_out/r04n_rt.log; tail -1 gpurun_out/r04n_rt.log
timeout 300 bash scripts/render_cli_configs.sh > gpurun_out/r04n_cli.log 2>&1; cat gpurun_out/r04n_cli.log
