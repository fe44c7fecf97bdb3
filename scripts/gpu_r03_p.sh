#!/bin/bash
# round 3, job P: a kernel change measured (batches / single frames on three workloads) and the whole GPU suite
mkdir -p gpurun_out
: > gpurun_out/r03p_variants.log
echo "== dragon" >> gpurun_out/r03p_variants.log; RT_VARIANTS="default,noshare,bpc3" timeout 300 python scripts/rt_variants.py 2>&1 | grep "{" >> gpurun_out/r03p_variants.log
echo "== statue depth 1" >> gpurun_out/r03p_variants.log; RT_VARIANTS="default,bpc3" timeout 300 python scripts/rt_variants.py statue.ply 1 2>&1 | grep "{" >> gpurun_out/r03p_variants.log
echo "== chessboard" >> gpurun_out/r03p_variants.log; RT_VARIANTS="default,bpc3" timeout 300 python scripts/rt_variants.py chessboard.tri 3 2>&1 | grep "{" >> gpurun_out/r03p_variants.log
cat gpurun_out/r03p_variants.log
timeout 1200 python -m pytest tests -m gpu -x -q --capture=sys > gpurun_out/r03p_pytest.log 2>&1
tail -5 gpurun_out/r03p_pytest.log | cut -c1-300
echo "== frame by frame" ; timeout 100 python scripts/raytrace_frame_by_frame.py 2>&1 | tail -8 | tee -a gpurun_out/r03p_variants.log
