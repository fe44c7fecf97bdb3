mkdir -p gpurun_out
export MI355_RS_SPLIT=1
(timeout 400 python -m pytest tests -m gpu -q -x --capture=sys -k "raster_pipeline or frame_overlap or async or batch" 2>&1 | tail -4) > gpurun_out/r04h_pytest.log; tail -3 gpurun_out/r04h_pytest.log
for d in 0 1; do
  echo "== MI355_NO_DIRECT_TURN=$d" >> gpurun_out/r04h.log
  MI355_NO_DIRECT_TURN=$d timeout 120 python scripts/raster_pipe_variants.py overlapped 2>&1 | tail -1 >> gpurun_out/r04h.log
  MI355_NO_DIRECT_TURN=$d timeout 120 python scripts/raytrace_frame_by_frame.py 2>&1 | grep -v amdgpu | tail -4 >> gpurun_out/r04h.log
done
cat gpurun_out/r04h.log
