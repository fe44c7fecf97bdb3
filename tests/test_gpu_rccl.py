"""The RCCL calls of the multi-GPU path on real hardware, as far as a one-GPU box allows: a one-rank "nccl" group runs
FrameGatherer's collective (the N>1 logic itself is covered by the world_size-2 gloo tests in test_multigpu_cpu.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_gather_through_rccl_with_one_rank():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    worker = os.path.join(os.path.dirname(__file__), "_rccl_one_rank_worker.py")
    p = subprocess.run([sys.executable, worker, str(29500 + os.getpid() % 2000)], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "RCCL_ONE_RANK_OK" in p.stdout, (p.stdout[-2000:], p.stderr[-4000:])
