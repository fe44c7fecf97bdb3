MI355_PIPE_DEBUG=1 timeout 100 python scripts/raster_pipe_variants.py overlapped 2>&1 | grep -v amdgpu
timeout 100 python -m pytest tests/test_gpu_raster_pipeline.py tests/test_gpu_frame_overlap.py -x -q 2>&1 | tail -1
