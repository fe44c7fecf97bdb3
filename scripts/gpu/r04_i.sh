mkdir -p gpurun_out
timeout 60 python scripts/gpu/r04_dbg.py > gpurun_out/r04i_dbg.log 2>&1; grep -v amdgpu.ids gpurun_out/r04i_dbg.log
(timeout 300 python -m pytest tests -m gpu -q -x --capture=sys -k "raster or frame_overlap or async or batch or screen_filling or span_buffers or mgpu" 2>&1 | tail -4) > gpurun_out/r04i_pytest.log; tail -3 gpurun_out/r04i_pytest.log
timeout 120 python scripts/raster_pipe_variants.py overlapped one_stream 2>&1 | tail -1 > gpurun_out/r04i.log
MI355_RS_SPLIT=1 timeout 120 python scripts/raster_pipe_variants.py overlapped one_stream 2>&1 | tail -1 >> gpurun_out/r04i.log
cat gpurun_out/r04i.log
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
(MI355_NO_OVERLAP=1 timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04i_prof -- python $R/scripts/raster_loop.py 6 200 2>&1 | tail -2) > $R/gpurun_out/r04i_prof.log
f=$(ls $R/gpurun_out/r04i_prof/*/*kernel_stats.csv | head -1); head -4 $f | cut -c1-160
