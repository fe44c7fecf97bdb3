#!/usr/bin/env python3
"""Timeline of the bench's overlapped raytrace launches from a rocprofv3 kernel trace: for every kernel of the timed region its
start and end (us, relative to the first), so that the gaps between a launch's selection kernel, its tracing kernel and the
copy-out show.  Run:  python scripts/rt_timeline.py [steps]   (runs bench.py under rocprofv3 --kernel-trace itself)"""
import csv, glob, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
steps = sys.argv[1] if len(sys.argv) > 1 else "40"
extra = sys.argv[2:]
with tempfile.TemporaryDirectory(dir="/tmp") as td:
    cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", td, "--", sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--warmup", "10",
           "--no-cpu-baseline", "--no-extra", "--no-pmc", "--repeats", "0"] + extra
    r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=300)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if line: print("bench ms_per_step", json.loads(line[-1]).get("ms_per_step"))
    rows = []
    for f in glob.glob(os.path.join(td, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"], row.get("Queue_Id", "?"), row.get("Grid_Size", "?"), row.get("Workgroup_Size", "?")))
rows.sort()
t0 = rows[0][0]
short = lambda n: "trace" if "k_raytrace" in n else ("select" if "k_tile_select" in n else ("copy" if "k_frames_copy" in n else ("fill" if "fillBuffer" in n else ("h2d" if "copyBuffer" in n else n.split("(")[0][-28:]))))
for r in rows:
    print("%12.1f %12.1f %9.1f q%s %s" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3], short(r[2])))
