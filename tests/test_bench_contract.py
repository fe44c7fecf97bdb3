"""bench.py's command-line contract where there is no GPU: the N > 1 launcher refuses loudly, and the committed dry-run line of the
N > 1 path (profiles/r05_bench_dryrun_n2.jsonl: `python bench.py --gpus 2 --dry-run` on the one-GPU box) has the shape the
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


@pytest.mark.parametrize("name,n", [("r05_bench_dryrun_n2.jsonl", 2), ("r05_bench_dryrun_n8.jsonl", 8), ("r06_bench_dryrun_n2.jsonl", 2), ("r06_bench_dryrun_n8.jsonl", 8)])
def test_the_dry_run_line_of_the_multi_gpu_path(name, n):
    """For every N the headline is the N = 1 workload -- dragon 1920x1080, 8 frames per GPU and step, bands over the GPUs, one exchange
    per step -- so that the driver's curve over N = 1, 2, 4, 8 compares like with like; BASELINE config 5
    (3840x2160) is a region of its own."""
    r = _line(name)
    n1_name = name[:3] + "_bench_n1.jsonl"
    n1 = _line(n1_name) if os.path.exists(os.path.join(ROOT, "profiles", n1_name)) else None
    assert r["n_gpus"] == n and r["dry_run"] is True and r["scaling"] == "weak"
    assert r["metric"] == "Mrays/sec" and r["unit"] == "Mrays/s" and r["value"] > 0 and r["higher_is_better"] is True
    assert "1920x1080" in r["config"]["workload"] and "dragon_vis.ply" in r["config"]["workload"]
    if n1:
        assert r["config"]["workload"] == n1["config"]["workload"] and r["metric"] == n1["metric"]
        assert abs(r["config"]["rays_per_frame"] - n1["config"]["rays_per_frame"]) / n1["config"]["rays_per_frame"] < 0.02   # same accounting
    assert r["config"]["frames_per_step"] == 8 * n
    mg = r["multi_gpu"]
    assert mg["transport"].startswith("dryrun") and mg["value_from"] in ("rank0", "spread")
    assert ("gather" in r["config"]["parallelism"]) == (mg["value_from"] == "rank0")
    for kind in ("rank0", "spread"):
        a = mg[kind]
        assert a["ms_per_step"] > 0 and a["exchange_ms"] > 0 and a["ingest_GBs"] > 0 and a["ingest_bytes_per_step"] > 0
        assert len(a["render_ms"]["per_rank"]) == n and 0 < a["render_ms"]["min"] <= a["render_ms"]["max"]
    # `value` is the faster assembly's
    assert mg[mg["value_from"]]["ms_per_step"] == min(mg["rank0"]["ms_per_step"], mg["spread"]["ms_per_step"])
    assert abs(r["ms_per_step"] - mg[mg["value_from"]]["ms_per_step"]) < 1e-3
    # rank 0 takes in the other ranks' bands of all 8 n frames; a rank of the spread assembly only those of the 8 frames it keeps
    rows_max = max(sum(1 for y in range(1080) if (y // 8) % n == k) for k in range(n))
    assert mg["rank0"]["ingest_bytes_per_step"] == (n - 1) * rows_max * 1920 * 4 * 8 * n
    assert mg["spread"]["ingest_bytes_per_step"] == (n - 1) * 8 * rows_max * 1920 * 4
    c5 = mg["config5"]
    assert "3840x2160" in c5["workload"] and c5["Mrays_per_s"] > 0 and c5["assembly"] in ("rank0", "spread")
    assert mg["whole_frames_no_exchange"]["frames_per_sec"] > 0
    if name.startswith("r06"):
        # round 6: the N > 1 line checks its own pictures -- every assembly that was timed hashes orbit frames f0 / f37 / f100 / f150 of the
        # assembled 1080p frames (and f37 / f100 of config 5's 3840x2160) against the reference's pins where they end up, the rasterizer
        # has a region of its own (chessboard, Phong and soft shadows, bands over the ranks, hashed against BASELINE config 2's pins),
        # and `value` is only printed because all of them agree
        sha = mg["assembled_sha"]
        assert mg["assembled_sha_ok"] is True and sha["pinned_frames"] == [0, 37, 100, 150]
        for kind in ("rank0", "spread"):
            assert sha[kind] == {"checked": 4, "differ": 0}
        assert sha["config5"]["checked"] == 2 and sha["config5"]["differ"] == 0 and sha["config5"]["pinned_frames"] == [37, 100]
        ras = mg["raster_1080p"]
        assert ras["assembly"] in ("rank0", "spread") and "%d frames of 1920x1080" % min(64, 8 * n) in ras["step"]
        for name_r, mode in (("phong", 6), ("softshadow", 8)):
            rr = ras[name_r]
            assert rr["mode"] == mode and rr["frames_per_sec"] > 0 and rr["assembled_sha"] == {"checked": 1, "differ": 0, "pinned_frames": [0]}
        assert mg["assembled_sha_checked"] == 4 + 4 + 2 + 1 + 1 and "invalid" not in r


def test_the_round_6_line_says_what_the_host_grants_and_which_rays_were_walked():
    """profiles/r06_bench_n1.jsonl: `cpu_baseline.cores` is what the box grants the process (its CFS quota, else its affinity mask), `threads`
    what was launched, with the CPU seconds per wall second the sample actually got, the share of periods it was throttled in and the
    burst figure of the probe beside the sustained value; `traced_Mrays_per_s` beside `value`; both fractions of the
    roofline block."""
    d = _line("r06_bench_n1.jsonl")
    c = d["cpu_baseline"]
    assert c["kind"] == "reference" and c["threads"] >= 1 and 1 <= c["cores"] <= c["host_threads"] and c["value"] > 0
    assert c["effective_cores_used"] > 0 and c["effective_cores_used"] <= c["threads"] + 0.5
    if c["cpu_quota_cores"] is not None:
        assert c["cores"] == min(c["cpu_quota_cores"], c["affinity_threads"]) and c["throttled_fraction"] is not None
        if c["throttled_fraction"] > 0.10:
            assert "throttled" in c["sample"]
    assert c["burst_Mrays_per_s"] is None or c["burst_Mrays_per_s"] > 0
    assert d["cpu_baseline_author"]["cores"] == c["cores"] and d["cpu_baseline_author"]["threads"] == c["threads"]
    assert 0 < d["traced_Mrays_per_s"] < d["value"] and d["config"]["traced_rays_per_frame"] < d["config"]["rays_per_frame"]
    assert "faces away" in d["config"]["rays_note"]
    r = d["roofline"]
    assert 0.0 < r["useful"]["frac"] < r["frac"] <= 1.0 and r["kernel_ms"] > 0 and r["traffic"] > 0
    recomputed = d["config"]["rays_per_frame"] * d["config"]["frames"] / (d["ms_per_step"] * 1e-3 * d["steps"]) / 1e6
    assert abs(recomputed - d["value"]) / d["value"] < 0.01


def test_the_committed_single_gpu_line_keeps_the_measurement_contract():
    """profiles/r05_bench_n1.jsonl (the round's evidence run) has what the contract asks of the N = 1 line: BASELINE.json's metric and
    unit on the configuration it is quoted on, a whole-job value that recomputes from the frames and the time, the roofline object with
    a fraction that does not exceed 1 and the kernel time it was measured on, the floor-of-work fraction beside the issue fraction,
    measured traffic, roofline blocks for the statue and 4K rows, and the CPU baselines: the reference's own code with the strict and
    with its author's flags, and configuration 1's single-thread rate."""
    path = os.path.join(ROOT, "profiles", "r05_bench_n1.jsonl")
    if not os.path.exists(path):
        pytest.skip("profiles/r05_bench_n1.jsonl has not been recorded yet")
    d = json.loads([l for l in open(path) if l.startswith("{")][-1])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["unit"] == "Mrays/s" and d["metric"].lower().startswith("mrays") and "mrays" in json.dumps(base).lower()
    assert d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f32" and d["scaling"] == "weak"
    assert "dragon_vis.ply" in d["config"]["workload"] and "1920x1080" in d["config"]["workload"] and "model" not in d["config"]
    frames = d["config"]["frames"]
    assert frames == d["steps"] * d["config"]["frames_per_step"] and d["config"]["frames_per_step"] == 8
    recomputed = d["config"]["rays_per_frame"] * frames / (d["ms_per_step"] * 1e-3 * d["steps"]) / 1e6
    assert abs(recomputed - d["value"]) / d["value"] < 0.01
    assert d["config"]["traced_rays_per_frame"] < d["config"]["rays_per_frame"]
    r = d["roofline"]
    assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["kernel_ms"] > 0 and r["traffic"] and r["traffic"] > 0
    assert 0.0 < r["timed_schedule"]["frac"] <= 1.0
    assert 0.0 < r["hbm"]["measured_frac"] < 1.0
    assert "this run" in r["counters"]["source"]                    # counters collected inside the run, not a stale file
    u = r["useful"]                                                 # the floor of the traversal's work, beside the issue fraction
    assert 0.0 < u["frac"] < r["frac"] and set(u["ops_per_record"]) == {"wide", "plane", "edge"}
    for row in ("statue_depth1_1080p", "dragon_4k"):
        rf = d["other_workloads"][row]["roofline"]
        assert 0.0 < rf["useful"]["frac"] < rf["frac"] <= 1.0 and rf["kernel_ms"] > 0 and rf["traffic"] > 0
    for name, rr in d["roofline_raster"].items():                    # the rasterizer: overlapped schedule and a frame by itself
        assert 0.0 < rr["frac"] < 1.0 and rr["gpu_ms_per_frame"] > 0 and "2000 frames" in rr["kernels"]
        assert 0.0 < rr["single_frame_frac"] <= rr["frac"] and rr["single_frame_gpu_ms"] >= rr["gpu_ms_per_frame"]
    c = d["cpu_baseline"]
    assert c["kind"] == "reference" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and "refcore_omp" in c["sample"]
    assert len(c["runs_Mrays_per_s"]) == 2
    a = d["cpu_baseline_author"]
    assert a["kind"] == "reference" and a["cores"] == c["cores"] and a["value"] > 0 and "refcore_omp_author" in a["sample"] and "-ffast-math" in a["sample"]
    assert d["cpu_baseline_port"]["kind"] == "port"
    c1 = d["cpu_baseline_config1"]
    assert c1["cores"] == 1 and c1["unit"] == "frames/s" and c1["value"] > 0 and "640x480" in c1["sample"]
    rows = d["other_workloads"]["render_cli_bench"]["rows"]
    assert len(rows) == 5 and all(x["fps_3_in_flight"] and x["fps_reference_loop"] for x in rows)
    assert set(d["other_workloads"]["shadowmap_1024_us"]) == {"chessboard.tri", "dragon_vis.ply"}


def _fake_cgroup(tmp_path, monkeypatch, proc_text, files):
    import bench
    root = tmp_path / "cg"
    for rel, text in files.items():
        f = root / rel
        f.parent.mkdir(parents=True, exist_ok=True)
        f.write_text(text)
    proc = tmp_path / "proc_cgroup"
    proc.write_text(proc_text)
    monkeypatch.setattr(bench, "CGROUP_FS", str(root))
    monkeypatch.setattr(bench, "PROC_CGROUP", str(proc))
    return bench


def test_cpu_grant_reads_the_quota_of_the_tightest_cgroup(tmp_path, monkeypatch):
    """bench.py's cpu_baseline.cores: cgroup v2 (cpu.max of the process's group or any group above it), cgroup v1 (cfs_quota / period),
    no quota (the affinity mask), and the throttling counters between two snapshots."""
    import os
    aff = len(os.sched_getaffinity(0))
    # v2, nested: the parent's 16-core quota is tighter than the leaf's "max"
    b = _fake_cgroup(tmp_path / "a", monkeypatch, "0::/pod/box\n",
                     {"cgroup.controllers": "cpu", "pod/box/cpu.max": "max 100000\n", "pod/cpu.max": "1600000 100000\n",
                      "pod/cpu.stat": "usage_usec 1\nnr_periods 100\nnr_throttled 40\nthrottled_usec 2000000\n"})
    g = b.cpu_grant()
    assert g["cpu_quota_cores"] == 16.0 and g["granted_cores"] == min(16.0, aff) and g["cpu_quota_from"].endswith("pod/cpu.max")
    s0 = b.cpu_stat()
    (tmp_path / "a" / "cg" / "pod" / "cpu.stat").write_text("usage_usec 1\nnr_periods 200\nnr_throttled 90\nthrottled_usec 5000000\n")
    s1 = b.cpu_stat(); s1["wall"] = s0["wall"] + 10.0
    d = b.cpu_stat_delta(s0, s1, 64)
    assert d["throttled_fraction"] == 0.5 and d["throttled_seconds_per_wall_second"] == 0.3 and d["threads"] == 64
    # v1: quota / period in the cpu controller's hierarchy
    b = _fake_cgroup(tmp_path / "b", monkeypatch, "3:cpu,cpuacct:/jobs/x\n2:memory:/y\n",
                     {"cpu/jobs/x/cpu.cfs_quota_us": "250000\n", "cpu/jobs/x/cpu.cfs_period_us": "100000\n",
                      "cpu/jobs/x/cpu.stat": "nr_periods 10\nnr_throttled 1\nthrottled_time 1000000\n"})
    g = b.cpu_grant()
    assert g["cpu_quota_cores"] == 2.5 and g["granted_cores"] == min(2.5, aff)
    assert b.cpu_stat()["throttled_usec"] == 1000.0            # (v1 counts nanoseconds)
    # no quota anywhere: the affinity mask is what the host grants; no throttling counters -> None, not 0
    b = _fake_cgroup(tmp_path / "c", monkeypatch, "0::/\n", {"cgroup.controllers": "cpu", "cpu.max": "max 100000\n"})
    g = b.cpu_grant()
    assert g["cpu_quota_cores"] is None and g["granted_cores"] == aff
    s0 = b.cpu_stat(); s1 = dict(s0, wall=s0["wall"] + 1.0)
    assert b.cpu_stat_delta(s0, s1, 8)["throttled_fraction"] is None
