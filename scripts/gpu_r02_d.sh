mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12) > gpurun_out/r02d_pytest.log
(timeout 300 python scripts/host_path_time.py 2>&1 | tail -3) > gpurun_out/r02d_host_path.log
cat gpurun_out/r02d_pytest.log gpurun_out/r02d_host_path.log
