mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_async.py tests/test_gpu_threads.py tests/test_gpu_lifecycle.py -x -q > gpurun_out/async_tests.log 2>&1; grep -v amdgpu.ids gpurun_out/async_tests.log | tail -5
timeout 300 python bench.py 2>/dev/null | tail -1 > gpurun_out/bench_try.json; python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_try.json").read())
print(d["value"], d["ms_per_step"]); print(d["seam"]); print(d["other_workloads"]); print({k: (v["frac"], v["gpu_ms_per_frame"]) for k, v in d["roofline_raster"].items()})
PY
