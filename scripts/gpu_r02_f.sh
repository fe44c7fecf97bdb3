mkdir -p gpurun_out
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
cat > /tmp/one.py <<'PY'
import sys; sys.path.insert(0, "/root/repo")
import renderer_amd as R
s = R.Scene(R.assets.mesh_path("dragon_vis.ply")); s.context()
for i in range(6): s.build_bvh_device()
PY
(timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02f_prof -- python /tmp/one.py 2>&1 | tail -3) > $R/gpurun_out/r02f_prof.log
cd $R
for f in $(find gpurun_out/r02f_prof -name "*kernel_stats.csv" | head -1); do cut -c1-60,200- $f | head -20; python - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-60s calls %5s total %10.1f us avg %9.2f us" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
PY
done
