mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
(timeout 100 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r04l_trace -- python $R/scripts/raster_loop.py 6 300 2>&1 | tail -2) > $R/gpurun_out/r04l_trace.log
f=$(ls $R/gpurun_out/r04l_trace/*/*kernel_trace.csv | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
q = collections.defaultdict(list)
for r in rows:
    q[r["Queue_Id"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:24]))
for k, v in q.items():
    v.sort()
    v = v[len(v)//2: len(v)//2 + 16]
    print("queue", k, len(q[k]), "kernels")
    prev = None
    for s, e, n in v:
        print("   %-24s dur %6.1f us  gap %6.1f us" % (n, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0)); prev = e
PY
