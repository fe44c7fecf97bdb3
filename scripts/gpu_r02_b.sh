mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_lifecycle.py tests/test_gpu_threads.py -m gpu -x -q 2>&1 | tail -6) > gpurun_out/r02b_pytest.log
(timeout 300 python scripts/raster_phases.py 2>&1 | tail -14) > gpurun_out/r02b_phases.log
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02b_prof -- python $R/scripts/raster_fps.py 2>&1 | tail -3) > $R/gpurun_out/r02b_prof.log
cd $R
for f in $(find gpurun_out/r02b_prof -name "*kernel_stats.csv" | head -1); do grep "k_rs_" $f | cut -c1-200; done
cat gpurun_out/r02b_pytest.log gpurun_out/r02b_phases.log; grep "mode2_fps" gpurun_out/r02b_prof.log
