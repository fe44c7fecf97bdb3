mkdir -p gpurun_out
MI355_PIPE_DEBUG=1 timeout 200 python scripts/raytrace_frame_by_frame.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/rt_fbf.txt
timeout 600 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_full.log 2>&1; tail -5 gpurun_out/pytest_full.log
