mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_frame_overlap.py tests/test_gpu_raster_pipeline.py -x -q 2>&1 | tail -15
