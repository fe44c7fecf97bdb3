mkdir -p gpurun_out
(timeout 900 python scripts/rt_sweep.py --profile --frames 4 --grid '[{"trav":0,"chunk":64,"bpc":1},{"trav":0,"chunk":64},{"trav":1,"chunk":64,"bpc":1},{"trav":1,"chunk":64}]' 2>&1 | tail -30) > gpurun_out/sweep5.log
