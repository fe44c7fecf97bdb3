mkdir -p gpurun_out
for t in '{}' '{"bpc":5}' '{"bpc":5,"nopipe":1}'; do
  timeout 200 python bench.py --no-extra --no-cpu-baseline --tune "$t" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['config']['tune'], 'Mrays/s', d['value'], 'ms/step', d['ms_per_step'], 'iso kernel_ms', r.get('kernel_ms'), 'frac', r.get('frac'), 'period', r.get('timed_region_ms_per_launch'))"
done | tee gpurun_out/waves5.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "tuning_knobs" 2>&1 | tail -2
