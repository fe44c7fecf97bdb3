# round 3, job D: the GPU suite after the multi-GPU fix, and the driver's bench invocation with the new line (counters in the run)
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12) > gpurun_out/r03d_pytest.log
tail -5 gpurun_out/r03d_pytest.log
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2> gpurun_out/r03d_bench.err | tail -1 > gpurun_out/r03d_bench.json ) 2>&1 | tail -3
tail -5 gpurun_out/r03d_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03d_bench.json"))
print({k: d[k] for k in ("value", "ms_per_step", "frames_per_sec")}, d.get("repeats"))
r = d["roofline"]
print({k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms")})
print(r.get("counters")); print(r.get("hbm")); print(r.get("l2")); print(r.get("reference_work_rate", {}).get("x_hbm_peak"))
ow = d.get("other_workloads", {})
print({k: ow.get(k) for k in ("statue_depth1_1080p", "dragon_4k", "frame_by_frame_fps", "chessboard_phong_1080p_fps", "error")})
print(d.get("seam")); print(d.get("cpu_baseline"))
PY
