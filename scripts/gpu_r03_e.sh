# round 3, job E: whole suite (multi-GPU steps, front-end), and where the HBM writes of the bench kernel come from (WRITE_SIZE per variant)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -12) > gpurun_out/r03e_pytest.log
tail -5 gpurun_out/r03e_pytest.log
cd /tmp && export TMPDIR=/tmp
for v in default noshare bpc3 bpc3noshare; do
  case $v in default) T='{}';; noshare) T='{"noshare": 1}';; bpc3) T='{"bpc": 3}';; bpc3noshare) T='{"bpc": 3, "noshare": 1}';; esac
  (timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r03e_w_$v -- python $R/bench.py --pmc-child --tune "$T" 2>&1 | tail -1) > $R/gpurun_out/r03e_w_$v.log
  (timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r03e_f_$v -- python $R/bench.py --pmc-child --tune "$T" 2>&1 | tail -1) > $R/gpurun_out/r03e_f_$v.log
done
cd $R
python - <<'PY'
import csv, glob, collections
for v in ("default", "noshare", "bpc3", "bpc3noshare"):
    for c in ("w", "f"):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for f in glob.glob("gpurun_out/r03e_%s_%s/**/*counter_collection.csv" % (c, v), recursive=True):
            for row in csv.DictReader(open(f)):
                if "k_raytrace" in row["Kernel_Name"]:
                    a = acc[(row["Kernel_Name"][:48], row["Counter_Name"])]; a[0] += float(row["Counter_Value"]); a[1] += 1
        print(v, {k: (round(x[0] / x[1] / 1024, 1), x[1]) for k, x in acc.items()}, "(MB per launch, launches)")
PY
