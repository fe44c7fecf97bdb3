mkdir -p gpurun_out
R=$(pwd)
timeout 200 python bench.py --no-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['config']['tune'], 'Mrays/s', d['value'], 'ms/step', d['ms_per_step'], 'iso kernel_ms', r.get('kernel_ms'), 'frac', r.get('frac'), 'period', r.get('timed_region_ms_per_launch'))"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_write
(timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra 2>&1 | tail -2) > $R/gpurun_out/prof_write.log
cd $R
python - <<'PY'
import csv, glob, collections
f = sorted(glob.glob("gpurun_out/prof_write/**/*counter_collection.csv", recursive=True))[-1]
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(f)):
    if "k_raytrace<false, false, true, 4, true, false>" in row["Kernel_Name"]:
        a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
for k, v in acc.items(): print(k, "per launch: %.1f MB" % (v[0] / v[1] / 1024), "launches", v[1])
PY
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q 2>&1 | tail -2
