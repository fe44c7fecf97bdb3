mkdir -p gpurun_out
(timeout 600 python -m pytest tests -m gpu -x -q -k "reference_frame_hashes or option_matrix or tuning or antialias or small_frames or ragged or stats_variant or band" 2>&1 | tail -5) > gpurun_out/pytest6.log
(timeout 900 python scripts/rt_sweep.py --profile --frames 4 --grid '[{}, {"order":1}, {"exact":1}, {"lmin":1}, {"lmin":16}, {"lmin":32}, {"lmin":63}, {"lmin":16,"xmin":24,"rmin":32}, {"lmin":16,"xmin":4,"rmin":8}, {"chunk":128}, {"chunk":256}, {"bpc":2}, {"bpc":3}]' 2>&1 | tail -30) > gpurun_out/sweep6.log
