mkdir -p gpurun_out
(timeout 900 python scripts/rt_sweep.py --profile --frames 6 --grid '[{}, {"nosplit":1}, {"rmin":64}, {"rmin":32}, {"chunk":64,"rmin":8}, {"xmin":24,"rmin":32}, {"lmin":16}, {"bpc":1}]' 2>&1 | tail -30) > gpurun_out/sweep15.log
