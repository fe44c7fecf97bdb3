mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q -k "raster or shadow or reference_frame_hashes or band or cxx" 2>&1 | tail -5) > gpurun_out/pytest17.log
(timeout 900 python bench.py --steps 100 --warmup 10 2>&1 | tail -1) > gpurun_out/bench17.log
