mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(timeout 300 python -m pytest tests -m gpu -q -rs 2>&1 | tail -12) > gpurun_out/pytest_full.log
timeout 100 python scripts/raytrace_frame_by_frame.py 2>&1 | grep mode | tee gpurun_out/rt_fbf_bpc.txt
cd /tmp && export TMPDIR=/tmp
(MI355_NO_OVERLAP=1 timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_raster -- python $R/scripts/raster_loop.py 6 200 2>&1 | tail -2) > $R/gpurun_out/prof_stats_raster.log
(timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_raster_overlapped -- python $R/scripts/raster_loop.py 6 200 2>&1 | tail -2) > $R/gpurun_out/prof_stats_raster_overlapped.log
cd $R; tail -2 gpurun_out/pytest_full.log; grep fps gpurun_out/prof_stats_raster.log gpurun_out/prof_stats_raster_overlapped.log
