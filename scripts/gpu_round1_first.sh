mkdir -p gpurun_out
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40) > gpurun_out/pytest.log
(timeout 600 python bench.py --steps 50 --warmup 5 2>&1 | tail -20) > gpurun_out/bench.log
cd /tmp && export TMPDIR=/tmp
(timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof1 -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>&1 | tail -5) > $GRAFT_REPO_ROOT/gpurun_out/rocprof.log
cd $GRAFT_REPO_ROOT; find gpurun_out/prof1 -name "*stats*" | head; nproc; lscpu | grep "Model name"
