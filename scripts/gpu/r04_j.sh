mkdir -p gpurun_out
timeout 60 python scripts/gpu/r04_dbg.py > gpurun_out/r04j_dbg.log 2>&1; grep -v amdgpu.ids gpurun_out/r04j_dbg.log | tail -3
timeout 120 python scripts/raster_pipe_variants.py overlapped one_stream 2>&1 | tail -1 > gpurun_out/r04j.log
cat gpurun_out/r04j.log
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
(MI355_NO_OVERLAP=1 timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04j_prof -- python $R/scripts/raster_loop.py 6 200 2>&1 | tail -2) > $R/gpurun_out/r04j_prof.log
f=$(ls $R/gpurun_out/r04j_prof/*/*kernel_stats.csv | head -1); head -3 $f | cut -c1-160
