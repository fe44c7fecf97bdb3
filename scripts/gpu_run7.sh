mkdir -p gpurun_out
(timeout 600 python -m pytest tests -m gpu -x -q -k "reference_frame_hashes or option_matrix or tuning or antialias or small_frames or ragged or stats_variant or band" 2>&1 | tail -5) > gpurun_out/pytest7.log
(timeout 900 python scripts/rt_sweep.py --profile --frames 4 --grid '[{}, {"nolds":1}, {"bpc":3}, {"bpc":2}, {"bpc":1}, {"bpc":2,"nolds":1}, {"bpc":2,"lmin":16}, {"bpc":2,"lmin":4}, {"bpc":2,"xmin":24,"rmin":32}, {"bpc":2,"xmin":6,"rmin":8}, {"bpc":2,"exact":1}]' 2>&1 | tail -30) > gpurun_out/sweep7.log
