export TMPDIR=/tmp; R=$PWD; mkdir -p gpurun_out; cd /tmp
for g in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "VALUBusy VALUUtilization SALUBusy"; do
  n=$(echo $g | cut -d' ' -f1)
  timeout 120 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $R/gpurun_out/smpmc_$n -- python $R/scripts/shadowmap_time.py > /dev/null 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/smpmc_*")):
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[-1]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_sm_" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        for name, v in c.items():
            n = len(v) // 3
            print(k, name, ["%.4g" % (sum(v[i*n:(i+1)*n]) / max(1, n)) for i in range(3)])
PY
