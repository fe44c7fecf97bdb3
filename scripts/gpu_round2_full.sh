# Round-2 evidence run: full GPU test suite, bench line, rocprofv3 kernel stats + counters of the raytrace and raster kernels,
# and the side measurements DESIGN.md quotes.  scripts/make_profiles.py r02 distils gpurun_out/ into profiles/.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(timeout 1800 python -m pytest tests -m gpu -q -rs 2>&1 | tail -12) > gpurun_out/pytest_full.log
(timeout 900 python bench.py 2>&1 | tail -3) > gpurun_out/bench_full.log
{
  echo "== scripts/host_path_time.py"; timeout 300 python scripts/host_path_time.py 2>&1 | tail -6
  echo "== scripts/bvh_build_times.py"; timeout 300 python scripts/bvh_build_times.py 2>&1 | grep "rep [15]"
  echo "== scripts/raster_fps.py"; timeout 300 python scripts/raster_fps.py 2>&1 | tail -4
  echo "== scripts/raster_pipe_variants.py"; timeout 300 python scripts/raster_pipe_variants.py 2>&1 | tail -1
  echo "== scripts/raytrace_frame_by_frame.py"; timeout 300 python scripts/raytrace_frame_by_frame.py 2>&1 | tail -4
  echo "== scripts/rt_lane_util.py"; timeout 300 python scripts/rt_lane_util.py 2>&1 | tail -6
  echo "== tests/test_gpu_cull_margin.py"; timeout 300 python -m pytest tests/test_gpu_cull_margin.py -q -s 2>&1 | grep -E "pairs|passed|failed"
  echo "== scripts/first_frame_time.py"; timeout 300 python scripts/first_frame_time.py 2>&1 | tail -6
} > gpurun_out/misc_full.log 2>&1
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra"
# kernel stats: launches one after the other (tune flag 32: what roofline.kernel_ms is measured on), and the default run whose
# consecutive launches overlap inside the library (each launch then takes longer than the period between launches)
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --tune '{"nopipe": 1}' 2>&1 | tail -3) > $R/gpurun_out/prof_stats.log
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_overlapped -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra 2>&1 | tail -3) > $R/gpurun_out/prof_stats_overlapped.log
(timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- $B 2>&1 | tail -2) > $R/gpurun_out/prof_fetch.log
(timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- $B 2>&1 | tail -2) > $R/gpurun_out/prof_write.log
(timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/prof_sq -- $B 2>&1 | tail -2) > $R/gpurun_out/prof_sq.log
(timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $R/gpurun_out/prof_cache -- $B 2>&1 | tail -2) > $R/gpurun_out/prof_cache.log
(timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES --output-format csv -d $R/gpurun_out/prof_valu1 -- $B 2>&1 | tail -2) > $R/gpurun_out/prof_valu1.log
(timeout 600 rocprofv3 --kernel-trace --pmc VALUBusy VALUUtilization SALUBusy --output-format csv -d $R/gpurun_out/prof_valu2 -- $B 2>&1 | tail -2) > $R/gpurun_out/prof_valu2.log
# rasterizer (mode 6, chessboard 1080p, single-frame launches): traffic and issue counters of the three tiled kernels
RL="python $R/scripts/raster_loop.py 6 30"
(timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_rs_fetch -- $RL 2>&1 | tail -2) > $R/gpurun_out/prof_rs_fetch.log
(timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_rs_write -- $RL 2>&1 | tail -2) > $R/gpurun_out/prof_rs_write.log
(timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/prof_rs_sq -- $RL 2>&1 | tail -2) > $R/gpurun_out/prof_rs_sq.log
(timeout 300 rocprofv3 --kernel-trace --pmc VALUBusy VALUUtilization SALUBusy --output-format csv -d $R/gpurun_out/prof_rs_valu -- $RL 2>&1 | tail -2) > $R/gpurun_out/prof_rs_valu.log
cd $R; tail -3 gpurun_out/pytest_full.log; tail -1 gpurun_out/bench_full.log | cut -c1-400; cat gpurun_out/misc_full.log
