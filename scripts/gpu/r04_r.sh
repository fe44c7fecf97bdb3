mkdir -p gpurun_out
(timeout 400 python -m pytest tests -m gpu -q -x --capture=sys -k "shadow or light_rotation or raster_modes or screen_filling or two_lights or hashes or raster_scratch" 2>&1 | tail -5) > gpurun_out/r04r_pytest.log; tail -4 gpurun_out/r04r_pytest.log
timeout 100 python scripts/shadowmap_time.py 2>&1 | grep "us per"
for v in renderer_amd/lib/variant_*.so; do echo $v; MI355_RENDER_SO=$PWD/$v timeout 100 python scripts/shadowmap_time.py 2>&1 | grep "us per"; done
export TMPDIR=/tmp; R=$PWD; cd /tmp
(timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04r_prof -- python $R/scripts/shadowmap_time.py 2>&1 | tail -4) > $R/gpurun_out/r04r_prof.log
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/r04r_prof/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].split("(")[0]
    per[n].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for n, v in per.items():
    k = len(v) // 3
    if k < 10: continue
    print(n, len(v), ["%.1f" % (sum(v[i*k+3:(i+1)*k]) / (k - 3) / 1e3) for i in range(3)])
PY
find gpurun_out/r04r_prof -name "*trace.csv" -delete
