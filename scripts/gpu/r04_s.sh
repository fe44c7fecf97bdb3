mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
(timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04s_prof -- python $R/scripts/shadowmap_time.py 2>&1 | tail -3) > $R/gpurun_out/r04s.log
f=$(ls $R/gpurun_out/r04s_prof/*/*kernel_stats.csv | head -1); head -5 $f | cut -c1-200
