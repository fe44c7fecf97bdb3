# round 3, job F: shadow map by row items + light updates in stream order, multi-GPU fix, front-end with rendering; timings
mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12) > gpurun_out/r03f_pytest.log
tail -5 gpurun_out/r03f_pytest.log
(timeout 120 python scripts/shadowmap_time.py; MI355_SM_LEGACY=1 timeout 120 python scripts/shadowmap_time.py) > gpurun_out/r03f_shadowmap.log 2>&1
grep "us per" gpurun_out/r03f_shadowmap.log
(RT_VARIANTS="default,sharemin16,bpc3,bpc3 noshare,bpc3 sharemin16,bpc3 sharemin4" timeout 300 python scripts/rt_variants.py 2>&1 | tail -8) > gpurun_out/r03f_variants.log
cat gpurun_out/r03f_variants.log
