# The short evidence run behind profiles/r02_* after the overlap work (the counter passes of scripts/gpu_round2_full.sh were not
# repeated: the kernels they describe did not change): GPU tests, the bench line, kernel stats of launches one after the other
# and of the default overlapped run, the two frame-by-frame measurements.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(timeout 400 python -m pytest tests -m gpu -q -rs 2>&1 | tail -12) > gpurun_out/pytest_full.log
(timeout 400 python bench.py 2>&1 | tail -3) > gpurun_out/bench_full.log
{
  echo "== scripts/raster_pipe_variants.py"; timeout 100 python scripts/raster_pipe_variants.py 2>&1 | tail -1
  echo "== scripts/raytrace_frame_by_frame.py"; timeout 100 python scripts/raytrace_frame_by_frame.py 2>&1 | tail -4
} > gpurun_out/misc_overlap.log 2>&1
cd /tmp && export TMPDIR=/tmp
(timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra --tune '{"nopipe": 1}' 2>&1 | tail -3) > $R/gpurun_out/prof_stats.log
(timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_overlapped -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>&1 | tail -3) > $R/gpurun_out/prof_stats_overlapped.log
cd $R; tail -3 gpurun_out/pytest_full.log; tail -1 gpurun_out/bench_full.log | cut -c1-300; cat gpurun_out/misc_overlap.log; tail -1 gpurun_out/prof_stats.log | cut -c1-200; tail -1 gpurun_out/prof_stats_overlapped.log | cut -c1-200
# the rasterizer's kernels (mode 6 chessboard 1080p, 200 frames one per call): each kernel by itself (no overlap) and as the
# default runs them (three frames in flight)
cd /tmp
(MI355_NO_OVERLAP=1 timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_raster -- python $R/scripts/raster_loop.py 6 200 2>&1 | tail -2) > $R/gpurun_out/prof_stats_raster.log
(timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_raster_overlapped -- python $R/scripts/raster_loop.py 6 200 2>&1 | tail -2) > $R/gpurun_out/prof_stats_raster_overlapped.log
cd $R; grep fps gpurun_out/prof_stats_raster.log gpurun_out/prof_stats_raster_overlapped.log
