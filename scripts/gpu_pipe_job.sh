mkdir -p gpurun_out
timeout 200 python scripts/single_frame_cost_order.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/cost_order.txt
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tile_cull.py tests/test_gpu_batch.py tests/test_gpu_async.py -x -q > gpurun_out/cost_tests.log 2>&1; tail -3 gpurun_out/cost_tests.log
timeout 200 python bench.py --no-extra 2>/dev/null | tail -1 | cut -c1-400
