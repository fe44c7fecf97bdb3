R=$(pwd); mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra > $R/gpurun_out/ks.log 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/ks/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print("%-60s calls %5s avg %9.2f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
tail -1 gpurun_out/ks.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['seam'] if 'seam' in d else '', d['roofline']['kernel_ms'])"
