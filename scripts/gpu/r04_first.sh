# round 4, first GPU call: GPU tests, the N > 1 path in dry-run mode (2 and 8 ranks on the one GPU), the default bench line
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -q -x --capture=sys 2>&1 | tail -15) > gpurun_out/r04a_pytest.log
tail -4 gpurun_out/r04a_pytest.log
(timeout 500 python bench.py --gpus 2 --dry-run --steps 8 --warmup 2 2>gpurun_out/r04a_dry2.err | tail -1) > gpurun_out/r04a_dry2.jsonl
tail -c 600 gpurun_out/r04a_dry2.jsonl; tail -5 gpurun_out/r04a_dry2.err
(timeout 500 python bench.py --gpus 8 --dry-run --steps 4 --warmup 1 --no-weak 2>gpurun_out/r04a_dry8.err | tail -1) > gpurun_out/r04a_dry8.jsonl
tail -c 400 gpurun_out/r04a_dry8.jsonl; tail -5 gpurun_out/r04a_dry8.err
(timeout 700 python bench.py 2>gpurun_out/r04a_bench.err | tail -1) > gpurun_out/r04a_bench.jsonl
tail -c 1500 gpurun_out/r04a_bench.jsonl; tail -5 gpurun_out/r04a_bench.err
