# kernel timeline of consecutive raster frames (chessboard 1080p mode 6): start / duration of each kernel, relative to the first
R=$(pwd); mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl -- python $R/scripts/raster_loop.py 6 60 > /dev/null 2>&1
cd $R
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/tl/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "k_rs_" in r["Kernel_Name"] or "k_frame_copy" in r["Kernel_Name"]][-24:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-14s start %7.1f us  dur %6.1f us  end %7.1f" % (r["Kernel_Name"].split("<")[0].replace("void ", ""), (s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3))
PY
