# Round-1 evidence run: full GPU test suite, bench line, rocprofv3 kernel stats + HBM traffic counters.
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/pytest_full.log
(timeout 900 python bench.py 2>&1 | tail -3) > gpurun_out/bench_full.log
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -3) > $R/gpurun_out/prof_stats.log
(timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra 2>&1 | tail -2) > $R/gpurun_out/prof_fetch.log
(timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra 2>&1 | tail -2) > $R/gpurun_out/prof_write.log
(timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/prof_sq -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra 2>&1 | tail -2) > $R/gpurun_out/prof_sq.log
(timeout 600 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $R/gpurun_out/prof_cache -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra 2>&1 | tail -2) > $R/gpurun_out/prof_cache.log
(timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES --output-format csv -d $R/gpurun_out/prof_valu1 -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra 2>&1 | tail -2) > $R/gpurun_out/prof_valu1.log
(timeout 600 rocprofv3 --kernel-trace --pmc VALUBusy VALUUtilization SALUBusy --output-format csv -d $R/gpurun_out/prof_valu2 -- python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra 2>&1 | tail -2) > $R/gpurun_out/prof_valu2.log
cd $R; find gpurun_out/prof_* -name "*.csv" | head -40
