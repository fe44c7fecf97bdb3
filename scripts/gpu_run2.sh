mkdir -p gpurun_out
(timeout 600 python -m pytest tests -m gpu -x -q -k "reference_frame_hashes or option_matrix or tuning or antialias" 2>&1 | tail -5) > gpurun_out/pytest2.log
(timeout 900 python scripts/rt_sweep.py --profile --grid '[{"trav":0},{"trav":1},{"trav":0,"xmin":4},{"trav":0,"xmin":32},{"trav":1,"xmin":4},{"trav":1,"xmin":32},{"trav":1,"xmin":48,"rmin":48},{"trav":0,"xmin":24,"rmin":32,"chunk":64},{"trav":1,"xmin":24,"rmin":32,"chunk":64},{"trav":1,"xmin":24,"rmin":32,"chunk":1024},{"trav":1,"bpc":1},{"trav":1,"bpc":2}]' 2>&1 | tail -30) > gpurun_out/sweep2.log
