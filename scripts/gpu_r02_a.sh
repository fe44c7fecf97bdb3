# Round-2 GPU run A: full GPU suite on the tiled rasterizer, raster frame rates, kernel times of the raster kernels.
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r02a_pytest.log
(timeout 300 python scripts/raster_fps.py 2>&1 | tail -3) > gpurun_out/r02a_raster_fps.log
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02a_prof -- python $R/scripts/raster_fps.py 2>&1 | tail -3) > $R/gpurun_out/r02a_prof.log
cd $R
(timeout 600 python bench.py 2>&1 | tail -3) > gpurun_out/r02a_bench.log
find gpurun_out/r02a_prof -name "*kernel_stats.csv" | head -3
for f in $(find gpurun_out/r02a_prof -name "*kernel_stats.csv" | head -1); do head -30 $f; done
cat gpurun_out/r02a_pytest.log gpurun_out/r02a_raster_fps.log gpurun_out/r02a_bench.log
