mkdir -p gpurun_out
timeout 60 python scripts/gpu/r04_dbg.py > gpurun_out/r04e_fused.log 2>&1; grep -v amdgpu.ids gpurun_out/r04e_fused.log
