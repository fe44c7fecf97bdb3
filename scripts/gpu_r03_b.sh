# round 3, job B: the kernel after the early-out, a test subset, the bounds, and the lane utilisation counters with / without work sharing
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
(timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_mgpu.py tests/test_gpu_frame_overlap.py tests/test_gpu_tile_cull.py -m gpu -x -q 2>&1 | tail -8) > gpurun_out/r03b_pytest.log
tail -3 gpurun_out/r03b_pytest.log
(RT_VARIANTS="default,noshare,sharemin16,sharemin4,noshadows,noshadows noshare,norefl" timeout 300 python scripts/rt_variants.py 2>&1 | tail -10) > gpurun_out/r03b_variants.log
cat gpurun_out/r03b_variants.log
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra"
(timeout 200 rocprofv3 --kernel-trace --pmc VALUBusy VALUUtilization SALUBusy --output-format csv -d $R/gpurun_out/r03b_pmc_share -- $B 2>&1 | tail -2) > $R/gpurun_out/r03b_pmc_share.log
(timeout 200 rocprofv3 --kernel-trace --pmc VALUBusy VALUUtilization SALUBusy --output-format csv -d $R/gpurun_out/r03b_pmc_noshare -- $B --tune '{"noshare": 1}' 2>&1 | tail -2) > $R/gpurun_out/r03b_pmc_noshare.log
(timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/r03b_pmc_share2 -- $B 2>&1 | tail -2) > $R/gpurun_out/r03b_pmc_share2.log
cd $R
python - <<'PY'
import csv, glob, collections
for kind in ("r03b_pmc_share", "r03b_pmc_noshare", "r03b_pmc_share2"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("gpurun_out/%s/**/*counter_collection.csv" % kind, recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_raytrace" in row["Kernel_Name"]:
                a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    print(kind, {k: round(v[0] / v[1], 3) for k, v in acc.items()}, {k: v[1] for k, v in acc.items()})
PY
