mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_cull_margin.py -q -s 2>&1 | grep -E "pairs|passed|failed|Error") > gpurun_out/r02g_margin.log
cat gpurun_out/r02g_margin.log
