# round 3, job G: the banded shadow map (keys in LDS): parity subset and timings of the three generations of kernels
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mgpu.py tests/test_gpu_frontend.py tests/test_gpu_raster_pipeline.py -m gpu -x -q -k "shadow or raster_modes or light or 3ds or hashes or steps or frontend or two_lights" 2>&1 | tail -8) > gpurun_out/r03g_pytest.log
tail -4 gpurun_out/r03g_pytest.log
(timeout 120 python scripts/shadowmap_time.py; MI355_SM_LEGACY=2 timeout 120 python scripts/shadowmap_time.py; MI355_SM_LEGACY=1 timeout 120 python scripts/shadowmap_time.py) > gpurun_out/r03g_shadowmap.log 2>&1
grep "us per" gpurun_out/r03g_shadowmap.log
