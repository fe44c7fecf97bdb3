mkdir -p gpurun_out
(timeout 300 python -m pytest tests -m gpu -q -x --capture=sys -k "raster or frame_overlap or batch or screen_filling or span_buffers or mgpu or hashes" 2>&1 | tail -4) > gpurun_out/r04k_pytest.log; tail -3 gpurun_out/r04k_pytest.log
timeout 300 python scripts/raster_split_sweep.py 2>&1 | grep -v amdgpu > gpurun_out/r04k_sweep.log; cat gpurun_out/r04k_sweep.log
