# VALU occupancy counters of the bench kernel (separate passes; kernel trace only)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 20 --warmup 2 --no-cpu-baseline --no-extra"
(timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES --output-format csv -d $R/gpurun_out/pmc_valu1 -- $B 2>&1 | tail -3) > $R/gpurun_out/pmc_valu1.log
(timeout 300 rocprofv3 --kernel-trace --pmc VALUBusy VALUUtilization SALUBusy --output-format csv -d $R/gpurun_out/pmc_valu2 -- $B 2>&1 | tail -3) > $R/gpurun_out/pmc_valu2.log
(timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/pmc_valu3 -- $B 2>&1 | tail -3) > $R/gpurun_out/pmc_valu3.log
cd $R; ls gpurun_out/pmc_valu*/*/ 2>/dev/null | head
