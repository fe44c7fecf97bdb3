mkdir -p gpurun_out
./scripts/ubench/apicost > gpurun_out/r04b_apicost.log 2>&1; cat gpurun_out/r04b_apicost.log
timeout 300 python scripts/raster_pipe_variants.py > gpurun_out/r04b_pipe.log 2>&1; tail -1 gpurun_out/r04b_pipe.log
timeout 200 python scripts/raster_phases.py > gpurun_out/r04b_phases.log 2>&1; head -12 gpurun_out/r04b_phases.log
