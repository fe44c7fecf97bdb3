mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
G='[{"trav":1,"chunk":64}]'
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TCP_[A-Z_0-9a-z]*\|TCC_[A-Z_0-9a-z]*\|GRBM_[A-Z_]*" | sort -u | tr '\n' ' ' > $R/gpurun_out/counters_list.txt
(timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/pmc1 -- python $R/scripts/rt_sweep.py --frames 3 --grid "$G" 2>&1 | tail -3) > $R/gpurun_out/pmc1.log
(timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAVES SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc2 -- python $R/scripts/rt_sweep.py --frames 3 --grid "$G" 2>&1 | tail -3) > $R/gpurun_out/pmc2.log
(timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $R/gpurun_out/pmc3 -- python $R/scripts/rt_sweep.py --frames 3 --grid "$G" 2>&1 | tail -3) > $R/gpurun_out/pmc3.log
(timeout 300 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $R/gpurun_out/pmc4 -- python $R/scripts/rt_sweep.py --frames 3 --grid "$G" 2>&1 | tail -3) > $R/gpurun_out/pmc4.log
(timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc5 -- python $R/scripts/rt_sweep.py --frames 3 --grid "$G" 2>&1 | tail -3) > $R/gpurun_out/pmc5.log
(timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc6 -- python $R/scripts/rt_sweep.py --frames 3 --grid "$G" 2>&1 | tail -3) > $R/gpurun_out/pmc6.log
cd $R; find gpurun_out/pmc* -name "*.csv" | head -20
