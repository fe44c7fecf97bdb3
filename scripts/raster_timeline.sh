R=$(pwd); mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -- python $R/scripts/raster_loop.py 6 60 > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-60:]
prev = None
for r in rows[-16:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-28s dur %6.1f us  gap before %6.1f us" % (r["Kernel_Name"][:28], (e - s) / 1e3, (s - prev) / 1e3 if prev else 0))
    prev = e
PY
