mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q -k "raster or shadow or reference_frame_hashes or band or cxx" 2>&1 | tail -3) > gpurun_out/pytest21.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_raster -- python $R/bench.py --mesh chessboard.tri --mode 8 --steps 60 --warmup 5 --no-cpu-baseline --no-extra 2>&1 | tail -1) > $R/gpurun_out/prof_raster.log
