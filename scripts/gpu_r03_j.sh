#!/bin/bash
# round 3, job J: the walk step's sections gathered (tune[7]): variants side by side, pixels compared with the default's

mkdir -p gpurun_out
export RT_VARIANTS="default,trimin4,trimin8,trimin12,trimin16,trimin24,trimin32,trimin64,trimin16 boxmin4,trimin16 boxmin8,trimin24 boxmin16,boxmin8,bpc3,bpc3 trimin16,bpc3 trimin8"
timeout 600 python scripts/rt_variants.py > gpurun_out/r03j_variants.log 2>&1
RT_VARIANTS="default,trimin8,trimin16,trimin32" timeout 300 python scripts/rt_variants.py statue.ply 1 > gpurun_out/r03j_variants_statue.log 2>&1
cat gpurun_out/r03j_variants.log gpurun_out/r03j_variants_statue.log
