mkdir -p gpurun_out
(timeout 300 python -m pytest tests -m gpu -q -rs 2>&1 | tail -12) > gpurun_out/pytest_full.log; tail -2 gpurun_out/pytest_full.log
