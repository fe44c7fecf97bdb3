mkdir -p gpurun_out
MI355_PIPE_DEBUG=1 timeout 300 python -m pytest tests/test_gpu_raster_pipeline.py -x -q -s > gpurun_out/pipe_tests.log 2>&1; grep -v amdgpu.ids gpurun_out/pipe_tests.log | tail -12
MI355_PIPE_DEBUG=1 timeout 100 python scripts/raster_pipe_variants.py 2>gpurun_out/pipe_variants.err | tee gpurun_out/pipe_variants.json; grep mi355 gpurun_out/pipe_variants.err
timeout 150 bash scripts/raster_timeline.sh > gpurun_out/pipe_timeline.txt 2>&1; tail -26 gpurun_out/pipe_timeline.txt
timeout 600 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_full.log 2>&1; tail -3 gpurun_out/pytest_full.log
