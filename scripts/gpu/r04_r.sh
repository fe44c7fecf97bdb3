mkdir -p gpurun_out
echo default > gpurun_out/r04r_sm.log; timeout 100 python scripts/shadowmap_time.py 2>&1 | grep "us per" >> gpurun_out/r04r_sm.log
for v in renderer_amd/lib/variant_*.so; do echo $v >> gpurun_out/r04r_sm.log; MI355_RENDER_SO=$PWD/$v timeout 100 python scripts/shadowmap_time.py 2>&1 | grep "us per" >> gpurun_out/r04r_sm.log; done
echo default >> gpurun_out/r04r_sm.log; timeout 100 python scripts/shadowmap_time.py 2>&1 | grep "us per" >> gpurun_out/r04r_sm.log
cat gpurun_out/r04r_sm.log
