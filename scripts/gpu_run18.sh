mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q -k "reference_frame_hashes or option_matrix or tuning or antialias or small_frames or ragged or stats_variant or band" 2>&1 | tail -5) > gpurun_out/pytest18.log
(timeout 900 python scripts/rt_sweep.py --frames 6 --grid '[{}, {"lmin":1}, {"lmin":4}, {"lmin":16}, {"bpc":1}, {"bpc":2}, {"exact":1}, {"xmin":12,"rmin":16}, {"xmin":48,"rmin":48}]' 2>&1 | tail -30) > gpurun_out/sweep18.log
(timeout 300 python scripts/rt_rows.py 2>&1 | tail -16) > gpurun_out/rows18.log
(timeout 300 python scripts/rt_waveprof.py 2>&1 | tail -12) > gpurun_out/waveprof18.log
