# the overlap work's quick check on a GPU box: the frames-in-flight tests, the frame-by-frame rates of the three ways the library
# can run consecutive raster frames, the same for raytraced frames
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_raster_pipeline.py tests/test_gpu_frame_overlap.py -x -q 2>&1 | tail -2
MI355_PIPE_DEBUG=1 timeout 100 python scripts/raster_pipe_variants.py 2>&1 | grep -v amdgpu | tee gpurun_out/pipe_variants.json
timeout 100 python scripts/raytrace_frame_by_frame.py 2>&1 | grep mode | tee gpurun_out/rt_fbf.txt
