#!/bin/bash
# round 3, job N: the sharing threshold on the three raytrace workloads (batches of 8 / single frames)
mkdir -p gpurun_out
: > gpurun_out/r03n_sharemin.log
export RT_VARIANTS="default,sharemin4,sharemin12,sharemin16,sharemin24,sharemin32"
echo "== dragon 1080p depth 3" >> gpurun_out/r03n_sharemin.log
timeout 300 python scripts/rt_variants.py 2>&1 | grep "{" >> gpurun_out/r03n_sharemin.log
echo "== statue 1080p depth 1" >> gpurun_out/r03n_sharemin.log
timeout 300 python scripts/rt_variants.py statue.ply 1 2>&1 | grep "{" >> gpurun_out/r03n_sharemin.log
echo "== chessboard 1080p depth 3" >> gpurun_out/r03n_sharemin.log
timeout 300 python scripts/rt_variants.py chessboard.tri 3 2>&1 | grep "{" >> gpurun_out/r03n_sharemin.log
cat gpurun_out/r03n_sharemin.log
