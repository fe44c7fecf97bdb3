mkdir -p gpurun_out
(timeout 600 python -m pytest tests -m gpu -q -x --capture=sys -k "raster or frame_overlap or pipeline or mgpu or parity or async or batch or lifecycle or threads or frontend" 2>&1 | tail -6) > gpurun_out/r04d_pytest.log; tail -4 gpurun_out/r04d_pytest.log
MI355_HOST_PROF=1 timeout 300 python scripts/raster_pipe_variants.py overlapped one_stream > gpurun_out/r04d_pipe.log 2>&1; grep -v amdgpu.ids gpurun_out/r04d_pipe.log | tail -12
MI355_RS_SPLIT=1 timeout 300 python scripts/raster_pipe_variants.py overlapped > gpurun_out/r04d_pipe_split.log 2>&1; grep -v amdgpu.ids gpurun_out/r04d_pipe_split.log | tail -2
