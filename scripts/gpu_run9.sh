mkdir -p gpurun_out
(timeout 600 python -m pytest tests -m gpu -x -q -k "reference_frame_hashes or option_matrix or tuning or antialias or small_frames or ragged or stats_variant or band" 2>&1 | tail -5) > gpurun_out/pytest9.log
(timeout 900 python scripts/rt_sweep.py --frames 6 --grid '[{}, {"nocoop":1}, {"bpc":2}, {"bpc":3}, {"lmin":16}, {"xmin":6,"rmin":8}, {"xmin":24,"rmin":32}, {"nolds":1}, {"exact":1}]' 2>&1 | tail -30) > gpurun_out/sweep9.log
