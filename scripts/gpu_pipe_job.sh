mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_frame_overlap.py tests/test_gpu_raster_pipeline.py tests/test_gpu_batch.py -x -q 2>&1 | tail -2
timeout 100 python scripts/raytrace_frame_by_frame.py 2>&1 | grep "mode 9" | tee gpurun_out/rt_fbf_final.txt
