#!/bin/bash
mkdir -p gpurun_out
tag=$1
for i in 1 2; do
  timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --capture=sys > gpurun_out/r03l_${tag}_$i.log 2>&1
  echo "run $i rc=$?"; tail -2 gpurun_out/r03l_${tag}_$i.log | cut -c1-200
done
grep -B 12 "Fatal Python" gpurun_out/r03l_${tag}_*.log | cut -c1-250 | head -60
