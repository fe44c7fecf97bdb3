# round 3, first GPU job: the whole GPU suite with the work sharing on by default (quad / sharing knobs included), then the variants
mkdir -p gpurun_out
(timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r03a_pytest.log
tail -4 gpurun_out/r03a_pytest.log
(timeout 400 python scripts/rt_variants.py 2>&1 | tail -20) > gpurun_out/r03a_variants.log
cat gpurun_out/r03a_variants.log
(timeout 200 python bench.py --no-cpu-baseline --no-extra --steps 100 --warmup 10 2>&1 | tail -1 | cut -c1-400) > gpurun_out/r03a_bench.log
cat gpurun_out/r03a_bench.log
