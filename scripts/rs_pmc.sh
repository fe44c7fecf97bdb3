# Hardware counters of the raster frame's three kernels (chessboard 1080p, frames one after the other: MI355_NO_OVERLAP=1); one rocprofv3
# --pmc pass per group; averages per launch.   bash scripts/rs_pmc.sh [mode] > gpurun_out/rs_pmc.json
export TMPDIR=/tmp MI355_NO_OVERLAP=1; R=$PWD; MODE=${1:-6}; mkdir -p gpurun_out; cd /tmp
i=0
for g in "VALUBusy VALUUtilization SALUBusy" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $R/gpurun_out/rspmc_$i -- python $R/scripts/raster_loop.py $MODE 60 > /dev/null 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob("gpurun_out/rspmc_*")):
    fs = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))
    if not fs: continue
    for r in csv.DictReader(open(fs[-1])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "k_rs_" in k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(json.dumps({k: {n: round(sum(v[10:]) / max(1, len(v[10:])), 2) for n, v in c.items()} for k, c in acc.items()}, indent=1))
PY
