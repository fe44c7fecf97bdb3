mkdir -p gpurun_out
(timeout 300 python scripts/rt_lane_util.py 2>&1 | tail -12) > gpurun_out/r02g_lanes.log
cat gpurun_out/r02g_lanes.log
