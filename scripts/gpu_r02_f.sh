mkdir -p gpurun_out
R=$(pwd)
(timeout 900 python -m pytest tests/test_gpu_bvh.py -x -q 2>&1 | tail -15) > gpurun_out/r02f_bvh.log
(timeout 300 python scripts/bvh_build_times.py 2>&1 | grep "rep [15]") > gpurun_out/r02f_times.log
cd /tmp && export TMPDIR=/tmp
cat > /tmp/one.py <<'PY'
import sys; sys.path.insert(0, "/root/repo")
import renderer_amd as R
s = R.Scene(R.assets.mesh_path("dragon_vis.ply")); s.context()
for i in range(6): s.build_bvh_device()
PY
(timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02f_prof -- python /tmp/one.py 2>&1 | tail -3) > $R/gpurun_out/r02f_prof.log
cd $R
cat gpurun_out/r02f_bvh.log gpurun_out/r02f_times.log
for f in $(find gpurun_out/r02f_prof -name "*kernel_stats.csv" | head -1); do python - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-60s calls %5s total %10.1f us avg %9.2f us" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3))
PY
done
