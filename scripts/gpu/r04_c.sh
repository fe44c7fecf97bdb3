mkdir -p gpurun_out
MI355_HOST_PROF=1 timeout 300 python scripts/raster_pipe_variants.py overlapped > gpurun_out/r04c_pipe.log 2>&1; grep -v amdgpu.ids gpurun_out/r04c_pipe.log | tail -14
