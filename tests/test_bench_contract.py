"""bench.py's command-line contract where there is no GPU: the N > 1 launcher refuses loudly, and the committed dry-run line of the
N > 1 path (profiles/r04_bench_dryrun_n2.jsonl: `python bench.py --gpus 2 --dry-run` on the one-GPU box) has the shape the
driver's N > 1 run will have."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_gpu_present(), reason="a GPU is present: the refusal path is for boxes without one")
@pytest.mark.parametrize("args", [["--gpus", "2"], ["--gpus", "8", "--dry-run"], []])
def test_bench_refuses_loudly_without_a_gpu(args):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "needs a GPU" in (p.stderr + p.stdout)
    assert not p.stdout.strip().startswith("{")          # no JSON line: nothing was measured


def test_bench_rejects_a_launcher_that_disagrees_with_gpus():
    """--gpus N with WORLD_SIZE != N in the environment (a launcher started with another --nproc-per-node) is an error, not a silent
    one-GPU run that prints n_gpus: 1."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'if world != args.gpus:' in src and "self_launch(args.gpus)" in src


def _line(name):
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        pytest.skip("%s has not been recorded yet" % name)
    rows = [json.loads(l) for l in open(path) if l.strip().startswith("{")]
    assert rows, "%s holds no JSON line" % name
    return rows[-1]


@pytest.mark.parametrize("name,n", [("r04_bench_dryrun_n2.jsonl", 2), ("r04_bench_dryrun_n8.jsonl", 8)])
def test_the_dry_run_line_of_the_multi_gpu_path(name, n):
    r = _line(name)
    assert r["n_gpus"] == n and r["dry_run"] is True and r["scaling"] == "strong"
    assert r["metric"] == "Mrays/sec" and r["unit"] == "Mrays/s" and r["value"] > 0 and r["higher_is_better"] is True
    assert "3840x2160" in r["config"]["workload"] and "dragon_vis.ply" in r["config"]["workload"]
    assert r["config"]["frames_per_step"] == 8
    assert "gather" in r["config"]["parallelism"] and "rank 0" in r["config"]["parallelism"]
    mg = r["multi_gpu"]
    assert mg["transport"].startswith("dryrun") and mg["value_from"] == "rank0"
    for kind in ("rank0", "spread"):
        a = mg[kind]
        assert a["ms_per_step"] > 0 and a["exchange_ms"] > 0 and a["ingest_GBs"] > 0 and a["ingest_bytes_per_step"] > 0
        assert len(a["render_ms"]["per_rank"]) == n and 0 < a["render_ms"]["min"] <= a["render_ms"]["max"]
    # rank 0 takes in the other ranks' bands of all 8 frames; a rank of the spread assembly only those of the frames it keeps
    rows_max = max(sum(1 for y in range(2160) if (y // 8) % n == k) for k in range(n))
    assert mg["rank0"]["ingest_bytes_per_step"] == (n - 1) * rows_max * 3840 * 4 * 8
    assert mg["spread"]["ingest_bytes_per_step"] == (n - 1) * (8 // n) * rows_max * 3840 * 4
    assert abs(r["ms_per_step"] - mg["rank0"]["ms_per_step"]) < 1e-3
