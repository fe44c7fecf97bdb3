# round 3, job I: config 5 played by one GPU rank after rank, quick test subset after the tile-select change
mkdir -p gpurun_out
(timeout 300 python -m pytest tests/test_gpu_tile_cull.py tests/test_gpu_mgpu.py tests/test_gpu_lifecycle.py -m gpu -x -q 2>&1 | tail -4) > gpurun_out/r03i_pytest.log
tail -2 gpurun_out/r03i_pytest.log
(timeout 400 python scripts/mgpu_4k_sim.py 2>&1 | grep "{") > gpurun_out/r03i_mgpu4k.log
cat gpurun_out/r03i_mgpu4k.log
