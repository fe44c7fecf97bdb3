mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_frame_overlap.py tests/test_gpu_raster_pipeline.py -x -q > gpurun_out/overlap_tests.log 2>&1; grep -v amdgpu.ids gpurun_out/overlap_tests.log | tail -30
