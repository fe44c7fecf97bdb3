mkdir -p gpurun_out
(timeout 900 python scripts/rt_sweep.py --profile --frames 3 --grid '[{}, {"bpc":2}, {"bpc":1}, {"order":1}]' 2>&1 | tail -30) > gpurun_out/sweep8.log
