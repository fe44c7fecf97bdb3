#!/bin/bash
# round 3, job K: builds of the library side by side: batches and single frames, then the whole GPU suite.
# Make the builds here first (the GPU box receives built files): for each variant, edit / build / copy renderer_amd/lib/libmi355render.so to
# renderer_amd/lib/exp/libmi355render_<name>.so, then: gpurun -- 'BUILDS="a b" bash scripts/gpu_r03_k.sh' (renderer_amd loads MI355_RENDER_SO).
mkdir -p gpurun_out
: > gpurun_out/r03k_variants.log
for b in $BUILDS; do
  echo "== build $b" >> gpurun_out/r03k_variants.log
  MI355_RENDER_SO=$PWD/renderer_amd/lib/exp/libmi355render_$b.so RT_VARIANTS="default,noshare,bpc3,sharemin16,sharemin4" timeout 300 python scripts/rt_variants.py 2>&1 | grep "{" >> gpurun_out/r03k_variants.log
  echo "== build $b statue depth 1" >> gpurun_out/r03k_variants.log
  MI355_RENDER_SO=$PWD/renderer_amd/lib/exp/libmi355render_$b.so RT_VARIANTS="default,bpc3" timeout 300 python scripts/rt_variants.py statue.ply 1 2>&1 | grep "{" >> gpurun_out/r03k_variants.log
done
cat gpurun_out/r03k_variants.log
timeout 1200 python -m pytest tests -m gpu -x -q --capture=sys > gpurun_out/r03k_pytest.log 2>&1
tail -5 gpurun_out/r03k_pytest.log | cut -c1-300
