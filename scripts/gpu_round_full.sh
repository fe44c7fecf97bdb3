# The evidence run behind profiles/<tag>_*  (one gpurun call:  gpurun --timeout 2400 -- 'bash scripts/gpu_round_full.sh r06'):
# GPU tests, the bench line (counters collected in the run), kernel stats of launches one after the other and of the default
# overlapped run, the rasterizer's and the shadow map's kernels, side measurements (variants, frame by frame, the seam, render_cli -b),
# counter passes of the bench kernel; scripts/make_profiles.py <tag> then copies what is tracked into profiles/.
TAG=${1:-r06}
mkdir -p gpurun_out
R=$(pwd)
(timeout 900 python -m pytest tests -m gpu -q -rs --capture=sys 2>&1 | tail -12) > gpurun_out/pytest_full.log
tail -3 gpurun_out/pytest_full.log
(timeout 600 python bench.py 2>gpurun_out/bench_full.err | tail -1) > gpurun_out/bench_full.log
tail -1 gpurun_out/bench_full.log | cut -c1-300
# the N > 1 script path on this one GPU (every rank on cuda:0, exchange staged over gloo): the lines check their own assembled frames
# against the reference pins (multi_gpu.assembled_sha_ok) and carry the rasterizer's region; their timings mean nothing
for n in 2 8; do
  (timeout 900 python bench.py --gpus $n --dry-run --steps 10 --warmup 2 2>gpurun_out/bench_dryrun_n$n.err | tail -1) > gpurun_out/bench_dryrun_n$n.log
  python -c "import json,sys; d=json.loads(open('gpurun_out/bench_dryrun_n$n.log').read()); mg=d['multi_gpu']; print('dry run N=$n: value', d['value'], 'assembled_sha_ok', mg['assembled_sha_ok'], 'frames checked', mg['assembled_sha_checked'], 'raster', {k: v.get('frames_per_sec') for k, v in mg['raster_1080p'].items() if isinstance(v, dict)})" 2>&1 | tail -1
done
{
  echo "== scripts/rt_variants.py (dragon 1080p depth 3 and statue depth 1: batches of 8, single frames, frame hashes; default / dark shadow rays walked / round 5 loop (leaves in the step for batches, tests inside the walk loop) / work sharing off / three waves / 16 frames per launch)"
  for v in "noskip -DRT_SKIP_DARK=0" "r05loop -DRT_SKIP_DARK=0 -DRT_DEFER_BATCH=0 -DRT_FLUSH_OUT=0"; do set -- $v; n=$1; shift; bash scripts/build_rt_variant.sh $n "$@" > /dev/null 2>&1; done
  timeout 400 python scripts/rt_variants.py default noskip r05loop 'default:RT_TUNE={"noshare":1}' 'default:RT_TUNE={"bpc":3}' 'default:RT_B=16' 2>&1 | grep variant
  echo "== scripts/rs_variants.py default (chessboard 1080p frame by frame / batches of 8 / single frames, shadow maps; frame hashes)"; timeout 200 python scripts/rs_variants.py default 2>&1 | grep variant
  echo "== scripts/rs_tilelog.py, scripts/sm_tilelog.py (RS_TILELOG build: when the blocks of a raster frame's kernels and of the shadow map's tile kernel start, pass their phases and end)"
  bash scripts/build_rs_variant.sh tilelog -DRS_TILELOG=1 > /dev/null 2>&1
  (export MI355_WAVELOG=1 MI355_RENDER_SO=$R/renderer_amd/lib/variant_tilelog.so; timeout 100 python scripts/rs_tilelog.py 6 2>&1 | grep "^{" | cut -c1-1800; timeout 100 python scripts/sm_tilelog.py 2>&1 | grep "^{" | cut -c1-900)
  echo "== scripts/rs_pmc.sh 6 (counters of the raster frame's kernels)"; timeout 400 bash scripts/rs_pmc.sh 6 2>&1 | tail -80; rm -rf gpurun_out/rspmc_*
  echo "== scripts/shadowmap_time.py"; timeout 100 python scripts/shadowmap_time.py 2>&1 | grep "us per"
  echo "== scripts/raster_phases.py (the tile kernel's phases on counting frames; frames/s by threads per tile)"; timeout 100 python scripts/raster_phases.py 2>&1 | tail -14
  echo "== scripts/raytrace_frame_by_frame.py"; timeout 100 python scripts/raytrace_frame_by_frame.py 2>&1 | tail -8
  echo "== scripts/render_cli_configs.sh (render_cli -b, BASELINE.json's five configurations)"; timeout 200 bash scripts/render_cli_configs.sh 2>&1
  echo "== scripts/keep_canvas_rate.py (mi355_opts::keep_canvas: the synchronous seam and three frames in flight with and without, render_cli -b [--keep-canvas])"; timeout 280 python scripts/keep_canvas_rate.py 1000 2>&1 | grep -v amdgpu.ids
  echo "== scripts/ubench/apicost (host cost of the HIP calls a frame makes)"; (hipcc -O2 --offload-arch=gfx950 -o /tmp/apicost scripts/ubench/apicost.hip && timeout 60 /tmp/apicost) 2>&1 | tail -14
} > gpurun_out/misc_full.log 2>&1
cat gpurun_out/misc_full.log
cd /tmp && export TMPDIR=/tmp
(timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra --no-pmc --repeats 0 --tune '{"nopipe": 1}' 2>&1 | tail -3) > $R/gpurun_out/prof_stats.log
(timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_overlapped -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra --no-pmc --repeats 0 2>&1 | tail -3) > $R/gpurun_out/prof_stats_overlapped.log
(MI355_NO_OVERLAP=1 timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_raster -- python $R/scripts/raster_loop.py 6 200 2>&1 | tail -2) > $R/gpurun_out/prof_stats_raster.log
(timeout 100 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_shadowmap -- python $R/scripts/shadowmap_time.py 2>&1 | tail -3) > $R/gpurun_out/prof_stats_shadowmap.log
# counters of the bench kernel: share / noshare / three-wave build (utilisation, instructions, HBM traffic); --pmc passes by themselves
for v in default noshare bpc3; do
  case $v in default) T='{}';; noshare) T='{"noshare": 1}';; bpc3) T='{"bpc": 3}';; esac
  for g in "VALUBusy VALUUtilization SALUBusy" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    n=$(echo $g | cut -d' ' -f1)
    (timeout 120 rocprofv3 --kernel-trace --pmc $g --output-format csv -d $R/gpurun_out/pmc_${v}_$n -- python $R/bench.py --pmc-child --tune "$T" 2>&1 | tail -1) > $R/gpurun_out/pmc_${v}_$n.log
  done
done
cd $R
echo "== scripts/rt_pmc.py (TA / TD / wait counters of the bench kernel)"; (timeout 200 python scripts/rt_pmc.py > gpurun_out/rt_pmc.json 2>&1; tail -40 gpurun_out/rt_pmc.json)
find gpurun_out -name "*kernel_trace.csv" -size +2M -delete
python scripts/make_profiles.py $TAG
