mkdir -p gpurun_out
export MI355_RS_SPLIT=1
for q in 4 8 16; do for n in 3 4 5 6 7; do
  if [ $q = 4 ] && [ $n != 3 ]; then continue; fi
  echo "== GPU_MAX_HW_QUEUES=$q MI355_PIPE_SETS=$n" >> gpurun_out/r04g.log
  GPU_MAX_HW_QUEUES=$q MI355_PIPE_SETS=$n MI355_PIPE_DEBUG=1 timeout 120 python scripts/raster_pipe_variants.py overlapped 2>&1 | grep -v amdgpu.ids | tail -2 >> gpurun_out/r04g.log
done; done
cat gpurun_out/r04g.log
