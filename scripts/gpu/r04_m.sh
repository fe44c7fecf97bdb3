mkdir -p gpurun_out
(timeout 300 python -m pytest tests/test_gpu_post.py tests/test_gpu_parity.py -m gpu -q -x --capture=sys -k "mlaa or strips" 2>&1 | tail -12) > gpurun_out/r04m_pytest.log; tail -12 gpurun_out/r04m_pytest.log
