mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
for v in ONEWAVE FENCEALL STATIC; do
  cp renderer_amd/lib/variant_$v.so renderer_amd/lib/libmi355render.so
  echo "== $v" >> gpurun_out/r04f.log
  timeout 120 python scripts/raster_pipe_variants.py overlapped one_stream 2>&1 | tail -1 >> gpurun_out/r04f.log
done
echo "== SPLIT (two kernels)" >> gpurun_out/r04f.log
MI355_RS_SPLIT=1 timeout 120 python scripts/raster_pipe_variants.py overlapped one_stream 2>&1 | tail -1 >> gpurun_out/r04f.log
cat gpurun_out/r04f.log
cd /tmp && export TMPDIR=/tmp
for v in ONEWAVE STATIC; do
  cp $R/renderer_amd/lib/variant_$v.so $R/renderer_amd/lib/libmi355render.so
  (MI355_NO_OVERLAP=1 timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04f_prof_$v -- python $R/scripts/raster_loop.py 6 200 2>&1 | tail -2) > $R/gpurun_out/r04f_prof_$v.log
  f=$(ls $R/gpurun_out/r04f_prof_$v/*/*kernel_stats.csv | head -1); echo "== $v"; head -6 $f | cut -c1-150
done
