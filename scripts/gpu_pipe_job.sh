timeout 250 python -m pytest tests/test_gpu_frame_overlap.py tests/test_gpu_raster_pipeline.py tests/test_gpu_mgpu.py tests/test_gpu_api_errors.py tests/test_gpu_async.py -x -q 2>&1 | tail -3
