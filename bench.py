#!/usr/bin/env python3
"""bench.py -- the reference's headline benchmark on MI355X.

N = 1 (BASELINE.json configs[3], the config the metric is quoted on): a "step" is one pass of the hot path over one batch
of input: `renderer -b -m 9 dragon_vis.ply` at 1920x1080 (BVH raytrace with shadow rays and 2 reflection bounces) for
the next 8 cameras of the reference's auto-spin orbit, rendered by ONE launch (mi355_render_batch_device: every frame's
pixels are those of a single-frame render), scene + BVH resident in HBM, frames left in HBM.  value = Mrays/s (a ray =
one BVH_IntersectTriangles call).  The same line carries the reference-shaped call (`seam`: one frame per
call into host memory), the rasterizer (`roofline_raster`, configs[1]) and the CPU baselines.

N > 1 (BASELINE.json configs[4]): dragon at 3840x2160, a step = 8 frames of the orbit whatever N is ("scaling": "strong"):
every rank renders its interleaved 8-scanline bands of the 8 frames in one launch and ONE RCCL gather per step
assembles them on rank 0 (north_star: "a single RCCL gather over xGMI"; `value` is this region's).  A second region in the
same invocation assembles every frame on the rank that keeps it instead (one all-to-all exchange per step, no funnel);
`multi_gpu` reports both: step period, the slowest / fastest rank's render-only time, the exchange alone, a rank's ingest
rate, and a short weak-scaling run at 1080p (8 whole frames per GPU and step, no exchange).

Usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--dry-run]
       `--gpus N` without WORLD_SIZE in the environment starts itself under torch.distributed.run, one rank per GPU (the
       driver's own torchrun command line works as before: then WORLD_SIZE is set and must equal N).
       `--dry-run` (N > 1): all N ranks on cuda:0, the exchange staged through host memory over gloo -- every line of the
       N > 1 script path runs on a one-GPU box; the line says "transport": "dryrun" and is no scaling measurement.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
L2_PEAK_GBS = 34500.0       # aggregate L2 bandwidth of the eight XCDs, same guide ("L2 (per XCD)")
# vector-ALU issue peak: 256 CUs x 4 SIMDs x 16 lanes, one lane-operation per lane and clock, at the 2.4 GHz the guide quotes
VALU_PEAK_TLANEOPS = 256 * 4 * 16 * 2.4e9 / 1e12
BENCH_KERNEL = "k_raytrace<false, false, true, 4, true"        # the batched four-wave build the headline launches use


def algorithmic_bytes(st, width, rows):
    """B = 32*N_pop + 36*N_tri + 48*N_plane + 96*N_hit + 4*W*H per raytraced frame, with the REFERENCE
    algorithm's counts (the counting build walks the tree in the reference's order)."""
    return 32 * st["node_pops"] + 36 * st["tri_tests"] + 48 * st["plane_pass"] + 96 * st["shaded_hits"] + 4 * width * rows


def own_bytes(st, width, rows):
    """The same sum for the walk the timed kernel does (counting build of the ORDERED walk, tune flag 8): 64 bytes per wide
    record fetched (both children's boxes; the counters hold two box tests per record and one for the virtual record above the
    root, which rides in the kernel arguments: one per ray), 32 per triangle block, 48 per edge record, 80 per shaded hit, 4 per
    pixel written."""
    rays = st["normal_rays"] + st["shadow_rays"]
    wide = max(0, st["node_pops"] - rays) // 2
    return 64 * wide + 32 * st["tri_tests"] + 48 * st["plane_pass"] + 80 * st["shaded_hits"] + 4 * width * rows


# The fewest vector instructions a lane can spend on a visited record (no FMA contraction: the reference's operations, each rounded):
#   a wide record = RayIntersectsBox for two children (Raytracer.cc:99-151): per child 3 packed subtractions + 3 packed multiplications
#     (both planes of an axis side by side), 3 min + 3 max, max3 + min3, 2 compares                                  = 16, x 2 = 32
#   the plane half of a triangle test (Raytracer.cc:245-267): o - centre (3), three dot products (5 each), d - n.o (1), the IEEE
#     division (10), 4 compares                                                                                        = 33
#   the edge half where it is reached (Raytracer.cc:269-297): hit = o + d s (6), three (dot product - d_i) (6 each), 3 compares,
#     squared distance (8), 1 compare                                                                                  = 36
# Shading, ray generation, the stack, the dispenser and the work sharing are NOT in it: it is the floor of the traversal itself.
USEFUL_OPS = {"wide": 32, "plane": 33, "edge": 36}


def useful_ops(st):
    """Lane-instructions of that floor for the records the ORDERED walk visits (its counting build, tune flag 8)."""
    rays = st["normal_rays"] + st["shadow_rays"]
    wide = max(0, st["node_pops"] - rays) // 2
    return USEFUL_OPS["wide"] * wide + USEFUL_OPS["plane"] * st["tri_tests"] + USEFUL_OPS["edge"] * st["plane_pass"]


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


CGROUP_FS = "/sys/fs/cgroup"            # (tests point these two at a fake tree)
PROC_CGROUP = "/proc/self/cgroup"


def _cgroup_dirs():
    """Directories whose CPU controller files bound this process, innermost first: cgroup v2 (unified) and v1 (cpu / cpuacct)."""
    v2, v1 = [], []
    txt = _read(PROC_CGROUP) or ""
    for line in txt.splitlines():
        parts = line.split(":", 2)
        if len(parts) != 3:
            continue
        hid, ctrl, path = parts
        if hid == "0" and ctrl == "":
            base = CGROUP_FS if os.path.exists(CGROUP_FS + "/cgroup.controllers") else CGROUP_FS + "/unified"
            p = path
            while True:
                v2.append(os.path.normpath(base + "/" + p))
                if p in ("", "/"):
                    break
                p = os.path.dirname(p)
        elif "cpu" in ctrl.split(","):
            p = path
            while True:
                v1.append(os.path.normpath(CGROUP_FS + "/cpu/" + p))
                if p in ("", "/"):
                    break
                p = os.path.dirname(p)
    return v2, v1


def cpu_grant():
    """What the host grants this process: hardware threads it sees, its affinity mask, and the CFS quota of the tightest
    cgroup above it (v2 cpu.max / v1 cpu.cfs_quota_us) in cores -- None where no quota is set or none is visible from inside."""
    ncpu = os.cpu_count() or 1
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = ncpu
    quota, where = None, None
    v2, v1 = _cgroup_dirs()
    for d in v2:
        t = _read(d + "/cpu.max")
        if t:
            a = t.split()
            if len(a) == 2 and a[0] != "max" and float(a[1]) > 0:
                q = float(a[0]) / float(a[1])
                if quota is None or q < quota:
                    quota, where = q, d + "/cpu.max"
    for d in v1:
        q_us, p_us = _read(d + "/cpu.cfs_quota_us"), _read(d + "/cpu.cfs_period_us")
        if q_us and p_us and float(q_us) > 0 and float(p_us) > 0:
            q = float(q_us) / float(p_us)
            if quota is None or q < quota:
                quota, where = q, d + "/cpu.cfs_quota_us"
    granted = min(aff, ncpu) if quota is None else min(float(aff), quota)
    return {"host_threads": ncpu, "affinity_threads": aff, "cpu_quota_cores": None if quota is None else round(quota, 2),
            "cpu_quota_from": where, "granted_cores": round(granted, 2)}


def cpu_stat():
    """Throttling counters of the cgroups above this process (summed over the levels that have them; the tightest level is the
    one that moves) and the CPU seconds this process and its finished children have used."""
    import resource
    out = {"nr_periods": 0, "nr_throttled": 0, "throttled_usec": 0.0, "seen": False}
    v2, v1 = _cgroup_dirs()
    for d in v2 + v1:
        t = _read(d + "/cpu.stat")
        if not t:
            continue
        kv = dict((a.split()[0], float(a.split()[1])) for a in t.splitlines() if len(a.split()) == 2)
        if "nr_periods" in kv:
            out["seen"] = True
            out["nr_periods"] += kv.get("nr_periods", 0)
            out["nr_throttled"] += kv.get("nr_throttled", 0)
            out["throttled_usec"] += kv.get("throttled_usec", kv.get("throttled_time", 0) / 1e3)
    ru_s, ru_c = resource.getrusage(resource.RUSAGE_SELF), resource.getrusage(resource.RUSAGE_CHILDREN)
    out["cpu_seconds"] = ru_s.ru_utime + ru_s.ru_stime + ru_c.ru_utime + ru_c.ru_stime
    out["wall"] = time.perf_counter()
    return out


def cpu_stat_delta(a, b, threads):
    """Between two cpu_stat() snapshots: the share of CFS periods in which the group was throttled, the time its threads were kept
    off the CPUs, and how many cores' worth of CPU time the sample actually got (CPU seconds / wall seconds)."""
    wall = max(b["wall"] - a["wall"], 1e-9)
    periods = b["nr_periods"] - a["nr_periods"]
    d = {"effective_cores_used": round((b["cpu_seconds"] - a["cpu_seconds"]) / wall, 2), "threads": threads}
    if a["seen"] and periods > 0:
        d["throttled_fraction"] = round((b["nr_throttled"] - a["nr_throttled"]) / periods, 3)
        d["throttled_seconds_per_wall_second"] = round((b["throttled_usec"] - a["throttled_usec"]) / 1e6 / wall, 2)
    else:
        d["throttled_fraction"] = None      # no CFS quota above this process (or its counters are not visible from inside)
    return d


def frame_pins(mesh, mode, w, h, depth):
    """{orbit frame: SHA-256 of its R,G,B bytes} for a configuration the reference's own output is pinned for: the survey's first
    frames (tests/golden/reference_pins.json) and the frames further along the orbit made from Raytracer.cc compiled in the build
    container (tests/golden/refcore_frame_pins.json).  Data files of the repository: nothing of the reference is read at run time."""
    pins = {}
    try:
        a = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_pins.json")))
        for f in a.get("frames", []):
            if (f["mesh"], f["mode"], f["w"], f["h"]) == (mesh, mode, w, h) and (mode < 9 or f["depth"] == depth):
                pins[0] = f["sha256"]
        b = json.load(open(os.path.join(ROOT, "tests", "golden", "refcore_frame_pins.json")))
        for f in b.get("frames", []):
            if (f["mesh"], f["mode"], f["w"], f["h"], f["depth"]) == (mesh, mode, w, h, depth) and not f.get("second_light"):
                pins[int(f["frame"])] = f["sha256"]
    except (OSError, ValueError, KeyError):
        pass
    return pins


def pmc_in_run(py_args, seconds=90):
    """Hardware counters of the bench kernel, measured NOW: this script again as a child under `rocprofv3 --pmc` (one pass per
    counter group: FETCH_SIZE and WRITE_SIZE do not fit one pass, MI355X_MICROARCH.md), a handful of the same launches, averages
    per launch of BENCH_KERNEL.  None when rocprofv3 is missing or a pass fails (the line then says so)."""
    import collections, csv, glob, shutil, signal, subprocess, tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, "rocprofv3 not found"
    groups = [["VALUBusy", "VALUUtilization", "SALUBusy"], ["FETCH_SIZE"], ["WRITE_SIZE"],
              ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "TCP_TCC_READ_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum"]]
    acc = collections.defaultdict(lambda: [0.0, 0])
    t_all = time.perf_counter()
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        for gi, g in enumerate(groups):
            out = os.path.join(td, "p%d" % gi)
            cmd = [exe, "--kernel-trace", "--pmc"] + g + ["--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__), "--pmc-child"] + py_args
            try:
                p = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
                try:
                    p.wait(timeout=seconds)
                except subprocess.TimeoutExpired:
                    os.killpg(p.pid, signal.SIGKILL)
                    return None, "counter pass %s timed out" % "+".join(g)
            except Exception as e:
                return None, "counter pass failed: %s" % e
            n = 0
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if BENCH_KERNEL in row["Kernel_Name"]:
                        a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1; n += 1
            if n == 0:
                return None, "counter pass %s produced no rows for the bench kernel" % "+".join(g)
    return {k: v[0] / v[1] for k, v in acc.items()}, "%d launches per counter, %.0f s" % (min(v[1] for v in acc.values()), time.perf_counter() - t_all)


def self_launch(n):
    """`python bench.py --gpus N` with no launcher around it: run the same command line under torch.distributed.run, one rank per
    GPU on this node (127.0.0.1 rendezvous on a free port); returns the launcher's exit code.  Rank 0's JSON line is the child's
    stdout, passed through."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=0, help="default 1920")
    ap.add_argument("--height", type=int, default=0, help="default 1080")
    ap.add_argument("--mesh", default="dragon_vis.ply")
    ap.add_argument("--mode", type=int, default=9)
    ap.add_argument("--depth", type=int, default=3, help="max_ray_depth of the raytrace modes (3 = the reference's default; 1 = primary + shadow rays, BASELINE configs[2])")
    ap.add_argument("--tune", default="{}", help="JSON dict of renderer_amd.tune() knobs")
    ap.add_argument("--frames-per-step", type=int, default=0,
                    help="frames of the orbit rendered by one launch per GPU (raytrace modes; mi355_render_batch_device, 1..64); "
                         "default 8 per GPU: with N GPUs a step is 8*N frames")
    ap.add_argument("--shard", choices=("auto", "frames", "bands"), default="auto",
                    help="N > 1: 'bands' (= auto) every GPU renders its interleaved screen bands (rows of 8x8 tiles) of every frame "
                         "of the step, one gather assembles the framebuffers (north_star, SURVEY 8e); 'frames' = every GPU renders "
                         "whole frames of the step (every N-th one)")
    ap.add_argument("--assemble", choices=("auto", "rank0", "spread"), default="auto",
                    help="N > 1, band sharding, the region `value` is taken from: 'rank0' = one gather per step onto rank 0 (north_star), "
                         "'spread' = frame j of a step is assembled on rank j %% N by one all-to-all exchange per step, 'auto' (default) = both are "
                         "timed, `value` is the faster one's and multi_gpu.value_from says which (with --one-assembly: rank0 only)")
    ap.add_argument("--one-assembly", action="store_true", help="N > 1: time only the --assemble region")
    ap.add_argument("--dry-run", action="store_true",
                    help="N > 1 on ONE GPU: every rank renders on cuda:0 and the exchange is staged through host memory over gloo; "
                         "runs the whole N > 1 path, measures nothing about scaling")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary (rasterizer) workloads")
    ap.add_argument("--no-cli", action="store_true", help="skip the render_cli -b runs of the five BASELINE configurations")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-weak", action="store_true", help="N > 1: skip the secondary regions (whole frames without an exchange; BASELINE config 5 at 3840x2160)")
    ap.add_argument("--no-pmc", action="store_true", help="do not collect the hardware counters of the bench kernel in this run")
    ap.add_argument("--pmc-child", action="store_true", help="(internal) a few launches of the bench workload and nothing else: run under rocprofv3 --pmc")
    ap.add_argument("--repeats", type=int, default=4, help="re-run the timed region this many more times for the spread (N = 1)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    import numpy as np
    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    n_dev = torch.cuda.device_count()
    if args.gpus > 1 and not args.dry_run and n_dev < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible on this box (use --dry-run to run the N > 1 path on one GPU)"
                         % (args.gpus, n_dev))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # one process per GPU: start ourselves under torch.distributed.run, exactly as the driver would
        sys.exit(self_launch(args.gpus))

    import torch.distributed as dist

    import renderer_amd as R
    from renderer_amd import multigpu

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the launcher's --nproc-per-node and --gpus must agree" % (args.gpus, world))
    dry = bool(args.dry_run and world > 1)
    if dry:
        local_rank = 0                       # every rank plays its part on the one GPU there is
        # N processes on one GPU oversubscribe its queues, and on this stack a kernel that needs scratch memory then now and again dies
        # of a memory fault (profiles/r06_analysis.md 10: 6 of 74 N = 8 dry runs while the rasterizer's batches took a build with scratch).
        # The dry run is a check of the script path, not of the schedule: its raytraced launches take the three-wave build (no scratch,
        # same pixels -- the line hashes them) unless --tune says otherwise.
        t_dry = dict(json.loads(args.tune))
        if "bpc" not in t_dry:
            t_dry["bpc"] = 3
            args.tune = json.dumps(t_dry)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    def all_reduce(values, op="sum"):
        """A few doubles reduced over the ranks (on the device with RCCL, on the host in a dry run)."""
        if world == 1:
            return [float(v) for v in values]
        t = torch.tensor(list(values), dtype=torch.float64, device="cpu" if dry else dev)
        dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN}[op])
        return [float(v) for v in t]

    def all_gather(value):
        if world == 1:
            return [float(value)]
        t = torch.tensor([float(value)], dtype=torch.float64, device="cpu" if dry else dev)
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [float(o[0]) for o in out]

    def barrier():
        if world > 1:
            dist.barrier()

    K, WU = args.steps, args.warmup
    # The headline workload is the same for every N: BASELINE's metric is quoted at 1920x1080 on 1 / 2 / 4 / 8 GPUs.  A step is
    # 8 frames of the orbit PER GPU (N = 1: 8, N = 8: 64): every GPU renders its interleaved screen bands of all 8 N frames in one
    # launch -- 8 frames' worth of rays per GPU and step whatever N is ("scaling": "weak") -- and one exchange per step assembles
    # the frames.  BASELINE configs[4] (3840x2160, bands over the GPUs) is a region of its own: multi_gpu.config5.
    W = args.width or 1920
    H = args.height or 1080
    B = max(1, min(64, args.frames_per_step if args.frames_per_step > 0 else 8 * world)) if args.mode >= 9 else 1
    scene = R.Scene(R.assets.mesh_path(args.mesh), device=local_rank)
    if args.mode >= 9:
        scene.bvh_update()                 # <mesh>.bvh cache in the scratch dir, else build (untimed, like -b)
    N_CAMS = 200                                  # the reference's benchmark orbit: frames f0..f199, then it repeats
    cams = [R.benchmark_frame(k) for k in range(N_CAMS)]

    def opts(**kw):
        o = R.default_opts(W, H, tune=json.loads(args.tune), max_ray_depth=args.depth, **kw)
        if world > 1 and not by_frames:
            o.band_rows, o.band_index, o.band_count, o.compact_rows = multigpu.BAND_ROWS, rank, world, 1
        return o

    by_frames = world > 1 and args.mode >= 9 and B % world == 0 and args.shard == "frames"
    if args.shard == "frames" and world > 1 and not by_frames:
        raise SystemExit("--shard frames needs a raytrace mode and frames-per-step divisible by the number of GPUs")
    B_local = B // world if by_frames else B          # frames per launch on this GPU
    # band sharding: where the frames of a step are put together (renderer_amd/multigpu.py).  The region `value` comes from is
    # north_star's single gather onto rank 0 unless --assemble spread; the other assembly is a second region (multi_gpu).
    can_spread = not by_frames and world > 1 and B > 1 and B % world == 0
    if args.assemble == "spread" and world > 1 and not can_spread:
        raise SystemExit("--assemble spread needs band sharding and frames-per-step divisible by the number of GPUs")

    def make_assembler(kind):
        if kind == "frames":
            return multigpu.BatchGatherer(W, H, dev, B_local)
        if kind == "spread":
            return multigpu.SpreadAssembler(W, H, dev, frames=B, staged=dry)
        return multigpu.FrameGatherer(W, H, dev, frames=B, staged=dry)

    primary = "frames" if by_frames else ("spread" if (can_spread and args.assemble == "spread") else "rank0")
    gather = make_assembler(primary)
    spread = primary == "spread"
    my_rows = gather.my_rows
    stream = torch.cuda.current_stream(dev)

    if args.mode in (7, 8):
        scene.shadowmap_render(0, cams[0][1][0])

    def frames_of_step(k):
        """Orbit frames this GPU renders in step k (all B of them in band mode, every world-th one in frame mode)."""
        fs = [(k * B + j) % N_CAMS for j in range(B)]
        return multigpu.frames_of_rank(fs, world, rank) if by_frames else fs

    def enqueue(k, o, slot, g=None, exchange=True):
        """One step: the next B frames of the orbit in one launch, then one exchange (N>1) through assembler g."""
        g = g or gather
        to_owner = isinstance(g, multigpu.SpreadAssembler)
        buf = g.send_buffer(slot)
        fs = frames_of_step(k)
        if B == 1:
            cam, lights, n = cams[fs[0]]
            scene.render_device(args.mode, cam, lights, n, o, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
        elif B_local == 1:
            cam, lights, n = cams[fs[0]]
            scene.render_device(args.mode, cam, lights, n, o, buf[0].data_ptr(), W * 4, 0, stream.cuda_stream)
        else:
            scene.render_batch_device(args.mode, [cams[f][0] for f in fs], [cams[f][1] for f in fs], cams[fs[0]][2], o,
                                      [buf[g.slot_of_frame(j) if to_owner else j].data_ptr() for j in range(B_local)], W * 4, None,
                                      stream.cuda_stream)
        if exchange:
            g.gather(slot)

    def timed_region(g, o, steps, warm):
        """`warm` untimed steps, then exactly `steps` steps between barrier + synchronize on both sides: wall seconds (max over
        ranks) and the launch stream's HIP-event milliseconds of this rank."""
        for k in range(warm):
            enqueue(k, o, k & 1, g)
        g.drain()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        e0.record(stream)
        for k in range(steps):
            enqueue(k, o, k & 1, g)
        e1.record(stream)
        g.drain()
        torch.cuda.synchronize(dev)
        barrier()
        torch.cuda.synchronize(dev)
        d = time.perf_counter() - t0
        return all_reduce([d], "max")[0], e0.elapsed_time(e1)

    def verify_assembled(g, kind, step_fn, frames_per_step, pins):
        """Outside every timed region: the steps that hold orbit frames with a reference pin are rendered and exchanged through
        assembler g, and the ASSEMBLED frames are hashed where they end up (rank 0 for a gather, rank j % N for the spread assembly).
        -> (frames checked, frames whose hash differs), the same on every rank."""
        import hashlib
        checked, bad = 0, 0
        for f, sha in sorted(pins.items()):
            k, j = divmod(f, frames_per_step)
            step_fn(k, 0)
            g.drain()
            torch.cuda.synchronize(dev)
            fr = g.frame(0)
            mine = None
            if kind == "spread":
                if j % world == rank:
                    mine = fr[j // world]
            elif rank == 0:
                mine = fr[j] if fr.dim() == 3 else fr
            if mine is not None:
                px = mine.contiguous().cpu().numpy().view(np.uint32)
                checked += 1
                bad += 0 if hashlib.sha256(R.rgb_bytes(px)).hexdigest() == sha else 1
        tot = all_reduce([checked, bad])
        return int(tot[0]), int(tot[1])

    if args.pmc_child:
        # launches one after the other, as the roofline's kernel_ms times them (the library does not overlap calls while a profiler
        # collects counters anyway)
        for k in range(12):
            enqueue(k, opts(), k & 1)
            torch.cuda.synchronize(dev)
        return

    # ---- untimed pre-pass: per-frame ray counts and algorithmic bytes from the counting kernel variant (which walks
    #      the tree in the reference's order: these ARE the reference algorithm's counts), once per orbit camera used
    o_stats = opts(collect_stats=1)
    t_own = dict(json.loads(args.tune)); t_own["profordered"] = 1
    o_own = R.default_opts(W, H, tune=t_own, collect_stats=1, max_ray_depth=args.depth)
    if world > 1 and not by_frames:
        o_own.band_rows, o_own.band_index, o_own.band_count, o_own.compact_rows = multigpu.BAND_ROWS, rank, world, 1
    used = sorted({f for k in range(K) for f in frames_of_step(k)})
    rays_f = np.zeros(N_CAMS, np.float64)
    culled_f = np.zeros(N_CAMS, np.float64)      # camera rays of tiles the production launch sets to black without tracing them
    abytes_f = np.zeros(N_CAMS, np.float64)
    obytes_f = np.zeros(N_CAMS, np.float64)      # the ordered walk's own bytes (raytrace modes)
    uops_f = np.zeros(N_CAMS, np.float64)        # ... and the floor of its vector instructions (useful_ops)
    scratch = torch.zeros((gather.max_rows, W), dtype=torch.int32, device=dev)
    for f in used:
        cam, lights, n = cams[f]
        scene.render_device(args.mode, cam, lights, n, o_stats, scratch.data_ptr(), W * 4, 0, stream.cuda_stream)
        torch.cuda.synchronize(dev)
        st = scene.fetch_stats().as_dict()
        rays_f[f] = st["normal_rays"] + st["shadow_rays"]
        abytes_f[f] = algorithmic_bytes(st, W, my_rows)
        if args.mode >= 9:
            scene.render_device(args.mode, cam, lights, n, o_own, scratch.data_ptr(), W * 4, 0, stream.cuda_stream)
            torch.cuda.synchronize(dev)
            st_own = scene.fetch_stats().as_dict()
            obytes_f[f] = own_bytes(st_own, W, my_rows)
            uops_f[f] = useful_ops(st_own)
            # (the production frame itself, one launch by itself: how many of its camera rays the tile culling never generates)
            t_c = dict(json.loads(args.tune)); t_c["nopipe"] = 1
            o_c = R.default_opts(W, H, tune=t_c, max_ray_depth=args.depth)
            o_c.band_rows, o_c.band_index, o_c.band_count, o_c.compact_rows = o_own.band_rows, o_own.band_index, o_own.band_count, o_own.compact_rows
            scene.render_device(args.mode, cam, lights, n, o_c, scratch.data_ptr(), W * 4, 0, stream.cuda_stream)
            torch.cuda.synchronize(dev)
            culled_f[f] = scene.culled_rays()
    my_rays = sum(rays_f[f] for k in range(K) for f in frames_of_step(k))
    my_abytes = sum(abytes_f[f] for k in range(K) for f in frames_of_step(k))
    my_culled = sum(culled_f[f] for k in range(K) for f in frames_of_step(k))
    total_rays, total_abytes, total_culled = all_reduce([my_rays, my_abytes, my_culled])

    # ---- W untimed warmup steps, then the timed region: exactly K steps, barrier + synchronize on both sides, max over ranks
    #      (gpu_ms: HIP events on the launch stream around the K launches)
    o_run = opts()
    dt, gpu_ms = timed_region(gather, o_run, K, WU)

    # ---- the same K-step region a few more times (N = 1): the spread of ms_per_step; `value` stays the first region's
    repeats = None
    if world == 1 and args.repeats > 0:
        rs = [dt * 1e3 / K]
        for _ in range(args.repeats):
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for k in range(K):
                enqueue(k, o_run, k & 1)
            gather.drain()
            torch.cuda.synchronize(dev)
            rs.append((time.perf_counter() - t1) * 1e3 / K)
        rs_sorted = sorted(rs)
        repeats = {"n": len(rs), "min": round(rs_sorted[0], 5), "median": round(rs_sorted[len(rs) // 2], 5), "max": round(rs_sorted[-1], 5),
                   "note": "ms_per_step of the timed region and of %d re-runs of the same %d steps; `value` and `ms_per_step` are the first region's" % (args.repeats, K)}

    # ---- the kernel by itself: consecutive launches of the timed region overlap inside the library (each on an internal
    #      stream; the tail of one launch -- a few waves finishing their tiles -- runs beside the head of the next), so no
    #      stream event brackets ONE of them.  The roofline's duration is therefore taken from the same launches one after
    #      the other on the launch stream (tune flag 32: the library does not overlap them), which is also what a
    #      rocprofv3 --kernel-trace of `bench.py --tune '{"nopipe": 1}'` shows per launch (profiles/).
    iso_ms = None
    if args.mode >= 9:
        t_iso = dict(json.loads(args.tune)); t_iso["nopipe"] = 1
        o_iso = R.default_opts(W, H, tune=t_iso, max_ray_depth=args.depth)
        if world > 1 and not by_frames:
            o_iso.band_rows, o_iso.band_index, o_iso.band_count, o_iso.compact_rows = multigpu.BAND_ROWS, rank, world, 1
        n_iso = max(5, min(K, 50))
        iso_buf = gather.send_buffer(0)

        def iso_launch(k):
            fs = frames_of_step(k)
            if B == 1:
                scene.render_device(args.mode, cams[fs[0]][0], cams[fs[0]][1], cams[fs[0]][2], o_iso, iso_buf.data_ptr(), W * 4, 0, stream.cuda_stream)
            elif B_local == 1:
                scene.render_device(args.mode, cams[fs[0]][0], cams[fs[0]][1], cams[fs[0]][2], o_iso, iso_buf[0].data_ptr(), W * 4, 0, stream.cuda_stream)
            else:
                scene.render_batch_device(args.mode, [cams[f][0] for f in fs], [cams[f][1] for f in fs], cams[fs[0]][2], o_iso,
                                          [iso_buf[j].data_ptr() for j in range(B_local)], W * 4, None, stream.cuda_stream)
        for k in range(3):
            iso_launch(k)
        torch.cuda.synchronize(dev)
        i0, i1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        i0.record(stream)
        for k in range(n_iso):
            iso_launch(k)
        i1.record(stream)
        torch.cuda.synchronize(dev)
        iso_ms = i0.elapsed_time(i1) / n_iso
        iso_abytes = sum(abytes_f[f] for k in range(n_iso) for f in frames_of_step(k)) / n_iso       # (this rank's launches)
        iso_obytes = sum(obytes_f[f] for k in range(n_iso) for f in frames_of_step(k)) / n_iso
        iso_uops = sum(uops_f[f] for k in range(n_iso) for f in frames_of_step(k)) / n_iso

    # sanity: the last assembled frame is a real picture
    if rank == 0:
        last = gather.frame((K - 1) & 1)
        nonblack = int((last != 0).sum().item())
        if B > 1:
            assert all(int((last[j] != 0).sum().item()) > 0 for j in range(last.shape[0])), "a frame of the last batch is empty"
        assert nonblack > 0, "rendered frame is empty"

    # ---- N > 1: what the step is made of, for both assemblies (north_star's gather onto rank 0, and every frame assembled on the
    #      rank that keeps it), and a weak-scaling run at 1080p
    mg = None
    if world > 1:
        mg = {"transport": "dryrun: %d ranks on ONE GPU (cuda:0), exchange staged through host memory over gloo -- exercises the N > 1 "
                           "script path, measures nothing about scaling; raytraced launches on the three-wave build (no scratch memory: N processes oversubscribe the one GPU's queues)" % world if dry else "rccl (torch.distributed backend nccl = RCCL over xGMI)",
              "sharding": ("whole frames: rank r renders every %d-th frame of a step" % world if by_frames else
                           "interleaved %d-scanline bands: band b -> rank b %% %d, compact [frames][rows][W] buffers" % (multigpu.BAND_ROWS, world)),
              "step": "%d frames of %dx%d" % (B, W, H)}

        def render_only(g):
            """The same launches into g's send buffers, no exchange: ms per step of every rank."""
            torch.cuda.synchronize(dev); barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n_r = max(4, K // 4)
            for k in range(2):
                enqueue(k, o_run, k & 1, g, exchange=False)
            torch.cuda.synchronize(dev)
            e0.record(stream)
            for k in range(n_r):
                enqueue(k, o_run, k & 1, g, exchange=False)
            e1.record(stream)
            torch.cuda.synchronize(dev)
            return all_gather(e0.elapsed_time(e1) / n_r)

        def exchange_only(g):
            """The exchange on already rendered buffers, one at a time: ms per step (slowest rank)."""
            barrier(); torch.cuda.synchronize(dev)
            n_g = max(4, K // 4)
            g.gather(0, async_op=False)
            torch.cuda.synchronize(dev); barrier()
            t1 = time.perf_counter()
            for k in range(n_g):
                g.gather(k & 1, async_op=False)
            torch.cuda.synchronize(dev)
            return all_reduce([(time.perf_counter() - t1) * 1e3 / n_g], "max")[0]

        def describe(g, kind, d_wall):
            per_rank = render_only(g)
            ex_ms = exchange_only(g)
            ingest = g.ingest_bytes_per_step() if hasattr(g, "ingest_bytes_per_step") else (world - 1) * g.max_rows * W * 4 * B_local
            return {"assembly": {"rank0": "one gather per step onto rank 0 (north_star: a single RCCL gather assembles the framebuffer)",
                                 "spread": "frame j of a step assembled on rank j % N: one grouped all-to-all exchange per step, no funnel",
                                 "frames": "whole frames gathered onto rank 0 in orbit order"}[kind],
                    "ms_per_step": round(d_wall * 1e3 / K, 5), "frames_per_sec": round(K * B / d_wall, 3), "Mrays_per_s": round(total_rays / d_wall / 1e6, 3),
                    "render_ms": {"min": round(min(per_rank), 4), "max": round(max(per_rank), 4), "per_rank": [round(v, 4) for v in per_rank],
                                  "note": "the same launches without the exchange, HIP events on every rank's launch stream"},
                    "exchange_ms": round(ex_ms, 4),
                    "ingest_bytes_per_step": int(ingest),
                    "ingest_GBs": round(ingest / (ex_ms * 1e-3) / 1e9, 2),
                    "ingest_note": ("what rank 0 takes in per step" if kind != "spread" else "what EVERY rank takes in per step") +
                                   ", over the standalone exchange's time; in the timed region the exchange of step k runs under the render of step k + 1 (two buffers)"}

        mg[primary] = describe(gather, primary, dt)
        mg["value_from"] = primary
        other = None if (by_frames or args.one_assembly) else ("spread" if primary == "rank0" else "rank0")
        dt_other = None
        if other == "spread" and not can_spread:
            mg["spread"] = {"skipped": "frames-per-step %d does not divide over %d ranks" % (B, world)}
            other = None
        if other:
            g2 = make_assembler(other)
            d2, _ = timed_region(g2, o_run, K, min(WU, 3))
            if rank == 0 or other == "spread":
                fr = g2.frame((K - 1) & 1)
                ok2 = all(int((fr[j] != 0).sum().item()) > 0 for j in range(fr.shape[0])) if fr is not None and fr.dim() == 3 else True
                assert ok2, "a frame of the second region's last step is empty"
            mg[other] = describe(g2, other, d2)
            dt_other = d2
        # ---- self-check (not timed): the assembled frames hash to the reference's pins, for every assembly that was timed
        pins_main = frame_pins(args.mesh, args.mode, W, H, args.depth)
        sha = {"pinned_frames": sorted(pins_main), "note": "orbit frames with a SHA-256 pin of the reference's own output (tests/golden/): rendered as part of their "
                                                          "step through each assembly and hashed where the assembled frame ends up; not in any timed region"}
        if pins_main:
            n_c, n_b = verify_assembled(gather, primary, lambda k, slot: enqueue(k, o_run, slot, gather), B, pins_main)
            sha[primary] = {"checked": n_c, "differ": n_b}
            if other:
                n_c2, n_b2 = verify_assembled(g2, other, lambda k, slot: enqueue(k, o_run, slot, g2), B, pins_main)
                sha[other] = {"checked": n_c2, "differ": n_b2}
        mg["assembled_sha"] = sha
        if other:
            del g2
        if dt_other is not None and args.assemble == "auto" and dt_other < dt:
            # `value` is the faster assembly's (both were timed the same way: K steps, barriers and synchronisation on both sides)
            mg["value_from"] = other
            dt = dt_other
            spread = other == "spread"
        mg["value_from_note"] = ("both assemblies timed, `value` from the faster" if dt_other is not None and args.assemble == "auto"
                                 else "only this assembly timed" if dt_other is None else "--assemble %s" % args.assemble)
        # for comparison: 8 whole frames per GPU and step, every rank keeps the frames it rendered (no sharding of a frame, no exchange)
        if not args.no_weak and args.mode >= 9:
            w_W, w_H = W, H
            wo = R.default_opts(w_W, w_H, tune=json.loads(args.tune))
            wbuf = [torch.zeros((w_H, w_W), dtype=torch.int32, device=dev) for _ in range(8)]
            def wstep(k):
                fs = [((k * world + rank) * 8 + j) % N_CAMS for j in range(8)]
                scene.render_batch_device(args.mode, [cams[f][0] for f in fs], [cams[f][1] for f in fs], cams[fs[0]][2], wo,
                                          [b.data_ptr() for b in wbuf], w_W * 4, None, stream.cuda_stream)
            for k in range(3):
                wstep(k)
            torch.cuda.synchronize(dev); barrier(); torch.cuda.synchronize(dev)
            n_w = max(10, K // 2)
            t1 = time.perf_counter()
            for k in range(n_w):
                wstep(k)
            torch.cuda.synchronize(dev); barrier(); torch.cuda.synchronize(dev)
            tw = all_reduce([time.perf_counter() - t1], "max")[0]
            mg["whole_frames_no_exchange"] = {"frames_per_step_per_gpu": 8, "steps": n_w, "frames_per_sec": round(n_w * 8 * world / tw, 2),
                                              "ms_per_step": round(tw * 1e3 / n_w, 4),
                                              "note": "every GPU renders 8 whole %dx%d frames of the orbit per step and keeps them: what N GPUs give when no frame "
                                                      "is sharded and nothing is exchanged -- the ceiling of the headline's curve" % (w_W, w_H)}
            del wbuf
        # BASELINE configs[4]: dragon 3840x2160, screen bands over the GPUs, one exchange per step; 8 frames per step whatever N is
        # (strong scaling), assembled the way `value` was
        if not args.no_weak and args.mode >= 9 and not by_frames:
            c_W, c_H, c_B = 3840, 2160, 8
            kind5 = mg["value_from"] if (mg["value_from"] != "spread" or c_B % world == 0) else "rank0"
            g5 = (multigpu.SpreadAssembler(c_W, c_H, dev, frames=c_B, staged=dry) if kind5 == "spread" else
                  multigpu.FrameGatherer(c_W, c_H, dev, frames=c_B, staged=dry))
            o5 = R.default_opts(c_W, c_H, tune=json.loads(args.tune))
            o5.band_rows, o5.band_index, o5.band_count, o5.compact_rows = multigpu.BAND_ROWS, rank, world, 1
            oc5 = R.default_opts(c_W, c_H, collect_stats=1)
            oc5.band_rows, oc5.band_index, oc5.band_count, oc5.compact_rows = multigpu.BAND_ROWS, rank, world, 1
            n5 = max(4, K // 2)
            used5 = sorted({(k * c_B + j) % N_CAMS for k in range(n5) for j in range(c_B)})
            rays5 = {}
            scr5 = torch.zeros((g5.max_rows, c_W), dtype=torch.int32, device=dev)
            for f in used5:
                scene.render_device(args.mode, cams[f][0], cams[f][1], cams[f][2], oc5, scr5.data_ptr(), c_W * 4, 0, stream.cuda_stream)
                torch.cuda.synchronize(dev)
                st5 = scene.fetch_stats()
                rays5[f] = st5.normal_rays + st5.shadow_rays
            del scr5
            def step5(k, slot):
                fs = [(k * c_B + j) % N_CAMS for j in range(c_B)]
                buf = g5.send_buffer(slot)
                scene.render_batch_device(args.mode, [cams[f][0] for f in fs], [cams[f][1] for f in fs], cams[fs[0]][2], o5,
                                          [buf[g5.slot_of_frame(j) if kind5 == "spread" else j].data_ptr() for j in range(c_B)], c_W * 4, None, stream.cuda_stream)
                g5.gather(slot)
            for k in range(2):
                step5(k, k & 1)
            g5.drain()
            barrier(); torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for k in range(n5):
                step5(k, k & 1)
            g5.drain()
            torch.cuda.synchronize(dev); barrier(); torch.cuda.synchronize(dev)
            t5 = all_reduce([time.perf_counter() - t1], "max")[0]
            r5 = all_reduce([float(sum(rays5[(k * c_B + j) % N_CAMS] for k in range(n5) for j in range(c_B)))])[0]
            pins5 = frame_pins(args.mesh, args.mode, c_W, c_H, 3)
            c5_checked, c5_bad = verify_assembled(g5, kind5, lambda k, slot: step5(k, slot), c_B, pins5) if pins5 else (0, 0)
            mg["assembled_sha"]["config5"] = {"checked": c5_checked, "differ": c5_bad, "pinned_frames": sorted(pins5)}
            mg["config5"] = {"workload": "%s, mode %d, %dx%d, screen bands x%d, %d frames per step (strong scaling: the step is the same for every N)"
                                         % (args.mesh, args.mode, c_W, c_H, world, c_B),
                             "assembly": kind5, "steps": n5, "ms_per_step": round(t5 * 1e3 / n5, 4), "frames_per_sec": round(n5 * c_B / t5, 2),
                             "Mrays_per_s": round(r5 / t5 / 1e6, 2), "rays_per_frame": round(r5 / (n5 * c_B), 1)}
            del g5
        # ---- the rasterizer at N > 1 (north_star: frames/s on chessboard.tri at 1 / 2 / 4 / 8 GPUs; the reference's frame loop is
        #      Rasterizers.cc:320-356 under renderer.cc:522-583): chessboard.tri at 1920x1080, per-pixel Phong (mode 6) and Phong + 3x3-PCF
        #      soft shadow map (mode 8); like the headline 8 frames per GPU and step, every GPU rasterizes its interleaved bands of all
        #      of them in one batched launch, ONE exchange per step assembles the frames; the first frame of the orbit is hashed
        #      against the reference's pins (BASELINE configs[1] and its soft-shadow variant) after assembly
        if not args.no_weak and not by_frames:
            try:
                r_W, r_H, r_B = 1920, 1080, min(64, 8 * world)
                kind_r = mg["value_from"] if (mg["value_from"] != "spread" or r_B % world == 0) else "rank0"
                g_r = (multigpu.SpreadAssembler(r_W, r_H, dev, frames=r_B, staged=dry) if kind_r == "spread" else
                       multigpu.FrameGatherer(r_W, r_H, dev, frames=r_B, staged=dry))
                trace = (lambda *a: print("[trace rank %d]" % rank, *a, file=sys.stderr, flush=True)) if os.environ.get("MI355_BENCH_TRACE") else (lambda *a: None)
                trace("raster region: scene")
                rsc = R.Scene(R.assets.mesh_path("chessboard.tri"), device=local_rank)
                rsc.shadowmap_render(0, cams[0][1][0])
                torch.cuda.synchronize(dev)
                trace("raster region: scene + shadow map done")
                ras = {"step": "%d frames of %dx%d (8 per GPU), bands x%d, one exchange per step" % (r_B, r_W, r_H, world), "assembly": kind_r}
                for r_mode, r_name in ((6, "phong"), (8, "softshadow")):
                    o_r = R.default_opts(r_W, r_H, tune=json.loads(args.tune))
                    o_r.band_rows, o_r.band_index, o_r.band_count, o_r.compact_rows = multigpu.BAND_ROWS, rank, world, 1

                    def rstep(k, slot, r_mode=r_mode, o_r=o_r):
                        fs = [(k * r_B + j) % N_CAMS for j in range(r_B)]
                        buf = g_r.send_buffer(slot)
                        trace("mode", r_mode, "step", k, "render")
                        rsc.render_batch_device(r_mode, [cams[f][0] for f in fs], [cams[f][1] for f in fs], cams[fs[0]][2], o_r,
                                                [buf[g_r.slot_of_frame(j) if kind_r == "spread" else j].data_ptr() for j in range(r_B)], r_W * 4, None,
                                                stream.cuda_stream)
                        if os.environ.get("MI355_BENCH_TRACE"):
                            torch.cuda.synchronize(dev)
                            trace("mode", r_mode, "step", k, "rendered; exchange")
                        g_r.gather(slot)
                    for k in range(3):
                        rstep(k, k & 1)
                    g_r.drain()
                    barrier(); torch.cuda.synchronize(dev)
                    n_r = max(20, K)
                    t1 = time.perf_counter()
                    for k in range(n_r):
                        rstep(k, k & 1)
                    g_r.drain()
                    torch.cuda.synchronize(dev); barrier(); torch.cuda.synchronize(dev)
                    t_r = all_reduce([time.perf_counter() - t1], "max")[0]
                    pins_r = frame_pins("chessboard.tri", r_mode, r_W, r_H, 3)
                    trace("mode", r_mode, "timed steps done; verify")
                    rc, rb = verify_assembled(g_r, kind_r, rstep, r_B, pins_r) if pins_r else (0, 0)
                    trace("mode", r_mode, "verified")
                    ras[r_name] = {"mode": r_mode, "steps": n_r, "ms_per_step": round(t_r * 1e3 / n_r, 4), "frames_per_sec": round(n_r * r_B / t_r, 1),
                                   "assembled_sha": {"checked": rc, "differ": rb, "pinned_frames": sorted(pins_r)}}
                    mg["assembled_sha"]["raster_" + r_name] = {"checked": rc, "differ": rb}
                mg["raster_1080p"] = ras
                trace("raster region: releasing")
                del g_r, rsc
                trace("raster region: released")
            except Exception as e:      # (a secondary region must not take the headline with it -- but its hashes, if it got that far, still count)
                mg["raster_1080p"] = {"error": str(e)}
        # one verdict over every assembly that was checked: a frame that differs anywhere voids the line's `value`
        checks = [v for k, v in mg["assembled_sha"].items() if isinstance(v, dict) and "differ" in v]
        mg["assembled_sha_ok"] = bool(checks) and all(v["differ"] == 0 for v in checks) and all(v["checked"] > 0 for v in checks)
        mg["assembled_sha_checked"] = int(sum(v["checked"] for v in checks))

    result = None
    if rank == 0:
        ms_per_step = dt * 1e3 / K
        period_ms = gpu_ms / K               # launch-stream time per step of the timed region (overlapped launches)
        kernel_ms = iso_ms if iso_ms else period_ms
        abytes_launch = iso_abytes if iso_ms else total_abytes / K / max(world, 1)
        result = {
            "metric": "Mrays/sec",
            "value": round(total_rays / dt / 1e6, 3),
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": K,
            "warmup": WU,
            "ms_per_step": round(ms_per_step, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic orbit: reference mesh %s (shipped asset), the reference's benchmark cameras f0..f199 (the orbit repeats), "
                    "BVH built on the GPU by the library" % args.mesh,
            "config": {"workload": "%s, BVH raytrace mode %d (primary + shadow rays + 2 reflection bounces), %dx%d, 1 light"
                                   % (args.mesh, args.mode, W, H),
                       "parallelism": ("single GPU" if world == 1 else
                                       "whole frames x%d (GPU r renders every %d-th frame of a step), 1 RCCL gather/step" % (world, world)
                                       if by_frames else
                                       "screen bands x%d, 1 RCCL all-to-all exchange/step (frame j assembled on rank j %% %d)" % (world, world) if spread
                                       else "screen bands x%d, 1 RCCL gather/step onto rank 0" % world) + (" [DRY RUN: all ranks on one GPU, gloo]" if dry else ""),
                       "frames_per_step": B, "frames": K * B,
                       "rays_per_frame": round(total_rays / (K * B), 1),
                       "traced_rays_per_frame": round((total_rays - total_culled) / (K * B), 1),
                       "rays_note": "rays_per_frame counts BVH_IntersectTriangles calls as the reference makes them (SURVEY 8d: a camera ray that misses "
                                    "everything counts, and the CPU baseline counts the same rays); traced_rays_per_frame leaves out the rays whose result "
                                    "cannot change a pixel and which the launch therefore does not walk: the camera rays of 8x8 tiles whose rays cannot reach "
                                    "any box of the tree's top (k_tile_select sets those pixels to black) and, since round 6, the shadow rays towards a light "
                                    "the hit faces away from (Raytracer.cc:472-475: `intensity < 0` adds nothing whether or not the ray is blocked)",
                       "tune": json.loads(args.tune)},
            "frames_per_sec": round(K * B / dt, 3),
            "traced_Mrays_per_s": round((total_rays - total_culled) / dt / 1e6, 3),
            "roofline": None,
        }
        if mg is not None and not mg.get("assembled_sha_ok", False) and mg.get("assembled_sha", {}).get("pinned_frames"):
            # (the frames the ranks assembled are NOT the reference's: no number is reported for a wrong picture)
            result["invalid"] = "assembled frames do not hash to the reference's pins (multi_gpu.assembled_sha): value withheld (would have been %.3f)" % result["value"]
            result["value"] = None
        if dry:
            result["dry_run"] = True
        if repeats:
            result["repeats"] = repeats
        # ---- roofline of the traversal kernel.  No dense contraction -> no MFMA; the scene is cache resident -> HBM does not
        #      bind either.  What binds is vector-ALU issue, so `frac` is the fraction of the chip's vector lanes
        #      doing useful work = VALUBusy x VALUUtilization, from counters collected in THIS run where rocprofv3 is present.
        ks = kernel_ms * 1e-3
        own_launch = iso_obytes if (iso_ms and args.mode >= 9) else None
        pmc, pmc_note, pmc_src = None, "not collected (N > 1, --no-pmc or a raster mode)", None
        if world == 1 and args.mode >= 9 and not args.no_pmc:
            child = ["--frames-per-step", str(B), "--mesh", args.mesh, "--mode", str(args.mode), "--tune", args.tune, "--width", str(W), "--height", str(H), "--depth", str(args.depth)]
            pmc, pmc_note = pmc_in_run(child)
            pmc_src = "this run: rocprofv3 --pmc passes of the same launches (%s)" % pmc_note if pmc else None
        if pmc is None and world == 1:
            tfile = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tfile):
                try:
                    tj = json.load(open(tfile))
                    pmc = tj.get("pmc_per_launch")
                    pmc_src = "STALE: profiles/traffic.json (committed passes of %s, another build of the kernel), because: %s" % (tj.get("round", "an earlier round"), pmc_note)
                except Exception:
                    pmc = None
        roof = {"kernel": "k_raytrace<STATS=false, EXACT_BOX=false, ORDERED=true, WAVES=2|3|4 by launch size, BATCH=frames>1> (ordered walk, work shared "
                          "inside a wave), preceded by k_tile_select (tiles no camera ray can hit anything in are set to black, ~1 % of the launch)",
                "kernel_ms": round(kernel_ms, 5),
                "kernel_ms_means": "one launch by itself: %d launches of the same steps one after the other on the launch stream (tune flag 32), "
                                   "HIP events around them" % (n_iso if iso_ms else K),
                "timed_region_ms_per_launch": round(period_ms, 5),
                "timed_region_note": "in the timed region consecutive launches overlap inside the library (internal streams, frames copied out "
                                     "on the launch stream): a launch every timed_region_ms_per_launch, each taking longer than kernel_ms",
                "frames_per_launch": B_local,
                "traffic": None}
        if pmc and "VALUBusy" in pmc:
            vb, vu = pmc["VALUBusy"] / 100.0, pmc.get("VALUUtilization", 0.0) / 100.0
            roof.update({"bound": "valu-issue",
                         "bound_note": "the contract's two bounds do not bind this kernel: there is no dense contraction (no MFMA) and the scene is cache "
                                       "resident (measured HBM traffic: see `hbm`); the vector ALUs' issue slots do.  achieved = lane-operations per second "
                                       "that do useful work = VALUBusy x VALUUtilization x peak; peak = 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz",
                         "achieved": round(vb * vu * VALU_PEAK_TLANEOPS, 3), "peak": round(VALU_PEAK_TLANEOPS, 3), "unit": "Tlane-op/s", "frac": round(vb * vu, 4),
                         "timed_schedule": (lambda vi: {"valu_busy": round(vi * 4.0 / (1024 * 2.4e9) / (ms_per_step * 1e-3), 4),
                                                       "frac": round(vi * 4.0 / (1024 * 2.4e9) / (ms_per_step * 1e-3) * vu, 4),
                                                       "note": "the same fraction for the schedule that is TIMED (launches overlapping three at a time): the launch's "
                                                               "vector wave-instructions x 4 cycles / 1024 SIMDs / 2.4 GHz over ms_per_step, x the active-lane share; "
                                                               "`frac` above is for one launch by itself (kernel_ms), which is how the counters are collected"})(pmc["SQ_INSTS_VALU"])
                                           if pmc.get("SQ_INSTS_VALU") else None,
                         "counters": {"source": pmc_src, "valu_busy_pct": round(pmc["VALUBusy"], 2), "valu_active_lanes_pct": round(pmc.get("VALUUtilization", 0.0), 2),
                                      "salu_busy_pct": round(pmc.get("SALUBusy", 0.0), 2),
                                      "valu_wave_instructions_per_launch": pmc.get("SQ_INSTS_VALU"), "salu_wave_instructions_per_launch": pmc.get("SQ_INSTS_SALU")}})
        else:
            roof.update({"bound": "hbm", "achieved": round((own_launch or abytes_launch) / ks / 1e9, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round((own_launch or abytes_launch) / ks / 1e9 / HBM_PEAK_GBS, 5),
                         "bound_note": "no hardware counters available (%s): the kernel's own algorithmic bytes over its time against the HBM peak" % pmc_note})
        if iso_ms and args.mode >= 9 and iso_uops > 0:
            roof["useful"] = {"lane_ops_per_launch": round(iso_uops, 1), "achieved": round(iso_uops / ks / 1e12, 3), "peak": round(VALU_PEAK_TLANEOPS, 3), "unit": "Tlane-op/s",
                              "frac": round(iso_uops / ks / 1e12 / VALU_PEAK_TLANEOPS, 4), "ops_per_record": USEFUL_OPS,
                              "means": "the FLOOR of the traversal's vector work over the kernel's time: the fewest instructions a lane can spend on the records the "
                                       "ordered walk visits (two slab tests per wide record, the plane half per triangle tested, the edge half where it is reached; "
                                       "counted on the same frames by the counting build of that walk) against the chip's lane-operations.  `frac` above is issue-slot "
                                       "occupancy; this one is work: the distance between the two is idle lanes of lockstep generations, the step's own bookkeeping "
                                       "(links, stack, masks, the next record's address), the work sharing and the shading between the walks"}
        hbm = {"peak": HBM_PEAK_GBS, "unit": "GB/s"}
        if own_launch:
            hbm.update({"own_bytes_per_launch": round(own_launch, 1), "achieved": round(own_launch / ks / 1e9, 3), "frac_own": round(own_launch / ks / 1e9 / HBM_PEAK_GBS, 5),
                        "own_bytes_mean": "the ordered walk's OWN algorithmic bytes: 64 B per wide record fetched, 32 per triangle block, 48 per edge record, 80 per "
                                          "shaded hit, 4 per pixel -- counted on the same frames by the counting build of that walk (tune flag 8; without tile "
                                          "culling and work sharing, which change the schedule, not the records a ray needs)"})
        if pmc and pmc.get("FETCH_SIZE") is not None and pmc.get("WRITE_SIZE") is not None:
            traffic = 2.0 * pmc["FETCH_SIZE"] * 1024.0 + pmc["WRITE_SIZE"] * 1024.0
            roof["traffic"] = round(traffic, 1)
            hbm.update({"measured_bytes_per_launch": round(traffic, 1), "measured_GBs": round(traffic / ks / 1e9, 2), "measured_frac": round(traffic / ks / 1e9 / HBM_PEAK_GBS, 5),
                        "measured_read_bytes": round(2.0 * pmc["FETCH_SIZE"] * 1024.0, 1), "measured_write_bytes": round(pmc["WRITE_SIZE"] * 1024.0, 1),
                        "measured_note": "FETCH_SIZE x 2 (the guide's gfx950 correction for 16-byte-per-lane loads) + WRITE_SIZE, KB -> bytes, separate passes; " + (pmc_src or "")})
        roof["hbm"] = hbm
        if pmc and pmc.get("TCP_TCC_READ_REQ_sum") is not None:
            l2b = pmc["TCP_TCC_READ_REQ_sum"] * 128.0
            roof["l2"] = {"read_requests_per_launch": round(pmc["TCP_TCC_READ_REQ_sum"], 1), "bytes_per_launch": round(l2b, 1), "GBs": round(l2b / ks / 1e9, 2),
                          "peak": L2_PEAK_GBS, "frac": round(l2b / ks / 1e9 / L2_PEAK_GBS, 5),
                          "hit_rate": round(pmc["TCC_HIT_sum"] / max(1.0, pmc["TCC_HIT_sum"] + pmc.get("TCC_MISS_sum", 0.0)), 4) if pmc.get("TCC_HIT_sum") is not None else None,
                          "note": "L1 -> L2 read requests (TCP_TCC_READ_REQ_sum) x 128-byte lines: an upper bound of the bytes the L2s deliver"}
        roof["reference_work_rate"] = {
            "bytes_per_launch": round(abytes_launch, 1), "GBs": round(abytes_launch / ks / 1e9, 3), "x_hbm_peak": round(abytes_launch / ks / 1e9 / HBM_PEAK_GBS, 4),
            "means": "SURVEY 8(d): the REFERENCE algorithm's bytes (its node pops / triangle tests / hits on the same frames, counted by the reference-order "
                     "build) over this kernel's time.  A work rate, not a bandwidth and not a roofline fraction: the timed kernel reaches the same pixels with "
                     "fewer visits (near child first, distance culling, tile culling), so it can exceed the HBM peak"}
        result["roofline"] = roof

    if rank == 0 and mg is not None:
        result["multi_gpu"] = mg

    # ---- secondary workloads (N=1 only, untimed by the driver's contract but reported)
    if rank == 0 and world == 1 and not args.no_extra:
        extra = {}
        try:
            chess = R.Scene(R.assets.mesh_path("chessboard.tri"), device=local_rank)
            buf = torch.zeros((H, W), dtype=torch.int32, device=dev)
            chess.shadowmap_render(0, cams[0][1][0])
            raster = {}
            for mode, name in ((6, "chessboard_phong_1080p"), (8, "chessboard_softshadow_1080p"), (2, "chessboard_points_1080p"), (3, "chessboard_wireframe_1080p")):
                o6 = R.default_opts(W, H)
                # (a fixed number of frames, whatever --steps says: three frames are in flight at a time, and a run of 20 frames -- the
                #  driver's --steps 20 in rounds 1-4 -- is a fifth ramp and drain: 21.5 k frames/s where 2 000 frames give 26 k)
                n_f = 400 if mode == 3 else 2000
                for k in range(5):
                    chess.render_device(mode, cams[k][0], cams[k][1], cams[k][2], o6, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
                torch.cuda.synchronize(dev)
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t1 = time.perf_counter()
                g0.record(stream)
                for k in range(n_f):
                    c_ = cams[k % N_CAMS]
                    chess.render_device(mode, c_[0], c_[1], c_[2], o6, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
                g1.record(stream)
                torch.cuda.synchronize(dev)
                extra[name + "_fps"] = round(n_f / (time.perf_counter() - t1), 2)
                chess.fetch_stats()                     # (raises if a frame was cut short by a bin overflow)
                if mode in (6, 8):
                    # B = 136 T + 8 W H (key clear) + 2*8 N_ztest + 4 W H (colour) [+ 36 N_shaded L for 3x3 PCF]
                    _, _, cst = chess.render(mode, cams[0][0], cams[0][1], cams[0][2], R.default_opts(W, H, collect_stats=1))
                    nbytes = 136 * chess.nt + 8 * W * H + 16 * cst.ztests + 4 * W * H + (36 * cst.plots if mode == 8 else 0)
                    ms = g0.elapsed_time(g1) / n_f
                    # ... and a frame by itself: the same frames with every kernel on the launch stream (tune flag 32), one after the other
                    o_alone = R.default_opts(W, H, tune={"nopipe": 1})
                    for k in range(5):
                        chess.render_device(mode, cams[k][0], cams[k][1], cams[k][2], o_alone, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
                    torch.cuda.synchronize(dev)
                    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a0.record(stream)
                    for k in range(400):
                        c_ = cams[k % N_CAMS]
                        chess.render_device(mode, c_[0], c_[1], c_[2], o_alone, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
                    a1.record(stream)
                    torch.cuda.synchronize(dev)
                    ms_alone = a0.elapsed_time(a1) / 400
                    raster[name] = {"bound": "hbm", "achieved": round(nbytes / (ms * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_frame": int(nbytes),
                                    "gpu_ms_per_frame": round(ms, 5),
                                    "single_frame_gpu_ms": round(ms_alone, 5), "single_frame_frac": round(nbytes / (ms_alone * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                    "single_frame_means": "400 frames with all of a frame's kernels on the launch stream, one frame after the other (no overlap "
                                                          "between frames): what ONE frame's kernels take; `frac` is the overlapped schedule's",
                                    "kernels": "k_rs_setup + k_rs_fill + k_rs_tile + k_frame_copy (consecutive frames overlap: "
                                    "each runs on one of three internal streams into a buffer of the library's, the launch stream copies it "
                                    "out; HIP events on the launch stream around %d frames)" % n_f,
                                    "ztests": int(cst.ztests), "shaded_pixels": int(cst.plots)}
            result["roofline_raster"] = raster
            # the same raster frames eight at a time (mi355_render_batch_device: side by side on internal streams)
            for mode, name in ((6, "chessboard_phong_1080p_batch8"), (8, "chessboard_softshadow_1080p_batch8")):
                o6 = R.default_opts(W, H)
                bufs8 = [torch.zeros((H, W), dtype=torch.int32, device=dev) for _ in range(8)]

                def raster_step(i):
                    fs = [(8 * i + j) % N_CAMS for j in range(8)]
                    chess.render_batch_device(mode, [cams[f][0] for f in fs], [cams[f][1] for f in fs], cams[fs[0]][2], o6,
                                              [b.data_ptr() for b in bufs8], W * 4, None, stream.cuda_stream)
                for i in range(3):
                    raster_step(i)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for i in range(100):
                    raster_step(i)
                torch.cuda.synchronize(dev)
                extra[name + "_fps"] = round(800 / (time.perf_counter() - t1), 2)
            # the headline workload frame by frame (one launch per frame: the latency-bound way to run the same frames)
            if args.mode >= 9:
                o1 = R.default_opts(W, H, tune=json.loads(args.tune), max_ray_depth=args.depth)
                for k in range(5):
                    scene.render_device(args.mode, cams[k][0], cams[k][1], cams[k][2], o1, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for k in range(200):
                    scene.render_device(args.mode, cams[k][0], cams[k][1], cams[k][2], o1, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
                torch.cuda.synchronize(dev)
                dt1 = time.perf_counter() - t1
                extra["frame_by_frame_fps"] = round(200 / dt1, 2)
                # the reference-shaped call: ONE frame per call into host memory (renderer.cc:522-583)
                seam = {}
                hb = [np.zeros((H, W), np.uint32) for _ in range(R.MAX_IN_FLIGHT)]
                kms = 0.0
                for k in range(5):
                    scene.render_into(args.mode, cams[k][0], cams[k][1], cams[k][2], o1, hb[0])
                t1 = time.perf_counter()
                for k in range(100):
                    kms += scene.render_into(args.mode, cams[k][0], cams[k][1], cams[k][2], o1, hb[0]).kernel_ms
                seam["sync_host_pageable_fps"] = round(100 / (time.perf_counter() - t1), 1)
                seam["single_frame_kernel_ms"] = round(kms / 100, 4)
                # ... and into a page-locked canvas, which is what the C++ host layer's Screen is since round 4 (Screen::_pixels is frame
                # memory of the library's, mi355_host_alloc): the kernels write the frame there themselves, no copy behind it
                pinned = [R.host_array((H, W)) for _ in range(R.MAX_IN_FLIGHT)]
                for k in range(5):
                    scene.render_into(args.mode, cams[k][0], cams[k][1], cams[k][2], o1, pinned[1])
                t1 = time.perf_counter()
                for k in range(200):
                    scene.render_into(args.mode, cams[k][0], cams[k][1], cams[k][2], o1, pinned[1])
                seam["sync_host_fps"] = round(200 / (time.perf_counter() - t1), 1)
                seam["single_frame_own_bytes_frac_of_hbm_peak"] = round(float(np.mean([obytes_f[k] for k in range(100) if obytes_f[k] > 0] or [0])) / (kms / 100 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                seam["single_frame_reference_work_x_hbm_peak"] = round(float(np.mean([abytes_f[k] for k in range(100) if abytes_f[k] > 0] or [0])) / (kms / 100 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                hb = pinned
                # (three in flight, not the four the API allows: measured 3 785 against 3 283 frames/s for raytraced 1080p frames -- four
                #  single-frame launches want more wave slots than a CU has, and four slot streams begin to share hardware queues)
                depth = min(3, R.MAX_IN_FLIGHT)
                try:
                    q = []
                    for k in range(8):
                        if len(q) == depth:
                            scene.render_wait(q.pop(0))
                        q.append(scene.render_async(args.mode, cams[k][0], cams[k][1], cams[k][2], o1, hb[k % R.MAX_IN_FLIGHT]))
                    for t in q:
                        scene.render_wait(t)
                    t1 = time.perf_counter(); q = []
                    for k in range(200):
                        if len(q) == depth:
                            scene.render_wait(q.pop(0))
                        q.append(scene.render_async(args.mode, cams[k][0], cams[k][1], cams[k][2], o1, hb[k % R.MAX_IN_FLIGHT]))
                    for t in q:
                        scene.render_wait(t)
                    seam["host_path_fps"] = round(200 / (time.perf_counter() - t1), 1)
                    # ... and the same two loops with mi355_opts::keep_canvas (the front-end's promise that only the render calls write into
                    # its canvases, Screen::_keepCanvas): a frame crosses PCIe only in the 8x8 tiles that are traced now or were in the
                    # canvas's last frame
                    ok = R.default_opts(W, H, max_ray_depth=o1.max_ray_depth, keep_canvas=1)
                    ok.tune[:] = list(o1.tune)
                    for k in range(5):
                        scene.render_into(args.mode, cams[k][0], cams[k][1], cams[k][2], ok, pinned[1])
                    t1 = time.perf_counter()
                    for k in range(200):
                        scene.render_into(args.mode, cams[k][0], cams[k][1], cams[k][2], ok, pinned[1])
                    seam["sync_host_keep_canvas_fps"] = round(200 / (time.perf_counter() - t1), 1)
                    t1 = time.perf_counter(); q = []
                    for k in range(200):
                        if len(q) == depth:
                            scene.render_wait(q.pop(0))
                        q.append(scene.render_async(args.mode, cams[k][0], cams[k][1], cams[k][2], ok, hb[k % depth]))
                    for t in q:
                        scene.render_wait(t)
                    seam["host_path_keep_canvas_fps"] = round(200 / (time.perf_counter() - t1), 1)
                finally:
                    for b in pinned:
                        R.host_array_free(b)
                seam["note"] = ("sync_host_fps: mi355_render, one synchronous frame per call into a page-locked canvas (the host layer's Screen): written by "
                                "the kernels themselves over PCIe, the background by the waves that have run out of pixels; sync_host_pageable_fps: the same "
                                "call into pageable memory (kernel + 8.3 MB D2H); "
                                "host_path_fps: the same frames through mi355_render_async / _wait, %d in flight on their own streams, into "
                                "frame memory of the library's (mi355_host_alloc); *_keep_canvas_fps: the two loops with mi355_opts::keep_canvas = 1 "
                                "(a canvas per frame in flight; written only where a frame can differ from the canvas's last); "
                                "single_frame_kernel_ms: hipEvent time of one frame's launch"
                                % depth)
                result["seam"] = seam
                extra["frame_by_frame_Mrays_per_s"] = round(float(sum(rays_f[k] for k in range(200) if rays_f[k] > 0)) / max(1, sum(1 for k in range(200) if rays_f[k] > 0)) * 200 / dt1 / 1e6, 1)
            # the other raytrace configurations of BASELINE.json on ONE GPU: configs[2] (statue, primary + shadow rays = depth 1) and
            # configs[4]'s frame size (dragon 3840x2160): 8 frames per launch like the headline, and the kernel time of one frame
            if args.mode >= 9:
                def rt_workload(mesh, w, h, depth):
                    sc = scene if mesh == args.mesh else R.Scene(R.assets.mesh_path(mesh), device=local_rank)
                    if sc is not scene:
                        sc.bvh_update()
                    ow = R.default_opts(w, h, max_ray_depth=depth, tune=json.loads(args.tune))
                    oc = R.default_opts(w, h, max_ray_depth=depth, collect_stats=1)
                    bufs = [torch.zeros((h, w), dtype=torch.int32, device=dev) for _ in range(8)]
                    n_cam = 40
                    rays = []
                    for f in range(n_cam):
                        sc.render_device(9, cams[f][0], cams[f][1], cams[f][2], oc, bufs[0].data_ptr(), w * 4, 0, stream.cuda_stream)
                        torch.cuda.synchronize(dev)
                        st = sc.fetch_stats()
                        rays.append(st.normal_rays + st.shadow_rays)
                    def step(i):
                        fs = [(8 * i + j) % n_cam for j in range(8)]
                        sc.render_batch_device(9, [cams[f][0] for f in fs], [cams[f][1] for f in fs], cams[fs[0]][2], ow, [b.data_ptr() for b in bufs], w * 4, None, stream.cuda_stream)
                    for i in range(5):
                        step(i)
                    torch.cuda.synchronize(dev)
                    n_s = 25
                    t1 = time.perf_counter()
                    for i in range(n_s):
                        step(i)
                    torch.cuda.synchronize(dev)
                    d = time.perf_counter() - t1
                    kms = [sc.render(9, cams[f][0], cams[f][1], cams[f][2], ow)[2].kernel_ms for f in range(0, n_cam, 4)]
                    row = {"Mrays_per_s": round(sum(rays) / n_cam * 8 * n_s / d / 1e6, 1), "frames_per_sec": round(8 * n_s / d, 1), "frames_per_launch": 8,
                           "rays_per_frame": round(sum(rays) / n_cam, 1), "single_frame_kernel_ms": round(float(np.mean(kms)), 4),
                           "workload": "%s, mode 9, max_ray_depth %d, %dx%d, orbit frames f0..f%d" % (mesh, depth, w, h, n_cam - 1)}
                    # the same roofline block as the headline's: one launch by itself (tune flag 32), the floor of its vector work from the
                    # counting build of the ordered walk, issue fraction and HBM traffic from counter passes of the same launches
                    try:
                        t_p = dict(json.loads(args.tune)); t_p["profordered"] = 1
                        op = R.default_opts(w, h, max_ray_depth=depth, tune=t_p, collect_stats=1)
                        uops = []
                        for f in range(0, n_cam, 4):
                            sc.render_device(9, cams[f][0], cams[f][1], cams[f][2], op, bufs[0].data_ptr(), w * 4, 0, stream.cuda_stream)
                            torch.cuda.synchronize(dev)
                            uops.append(useful_ops(sc.fetch_stats().as_dict()))
                        t_i = dict(json.loads(args.tune)); t_i["nopipe"] = 1
                        oi = R.default_opts(w, h, max_ray_depth=depth, tune=t_i)
                        def istep(i):
                            fs = [(8 * i + j) % n_cam for j in range(8)]
                            sc.render_batch_device(9, [cams[f][0] for f in fs], [cams[f][1] for f in fs], cams[fs[0]][2], oi, [b.data_ptr() for b in bufs], w * 4, None, stream.cuda_stream)
                        for i in range(2):
                            istep(i)
                        torch.cuda.synchronize(dev)
                        i0, i1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        i0.record(stream)
                        for i in range(10):
                            istep(i)
                        i1.record(stream)
                        torch.cuda.synchronize(dev)
                        k_ms = i0.elapsed_time(i1) / 10
                        ks_ = k_ms * 1e-3
                        u_launch = float(np.mean(uops)) * 8
                        rf = {"kernel_ms": round(k_ms, 5), "frames_per_launch": 8,
                              "useful": {"lane_ops_per_launch": round(u_launch, 1), "frac": round(u_launch / ks_ / 1e12 / VALU_PEAK_TLANEOPS, 4), "peak": round(VALU_PEAK_TLANEOPS, 3), "unit": "Tlane-op/s"}}
                        if not args.no_pmc:
                            pm, note = pmc_in_run(["--frames-per-step", "8", "--mesh", mesh, "--mode", "9", "--tune", args.tune, "--width", str(w), "--height", str(h), "--depth", str(depth)])
                            if pm and "VALUBusy" in pm:
                                vb_, vu_ = pm["VALUBusy"] / 100.0, pm.get("VALUUtilization", 0.0) / 100.0
                                rf.update({"bound": "valu-issue", "frac": round(vb_ * vu_, 4), "achieved": round(vb_ * vu_ * VALU_PEAK_TLANEOPS, 3), "peak": round(VALU_PEAK_TLANEOPS, 3),
                                           "unit": "Tlane-op/s", "valu_busy_pct": round(pm["VALUBusy"], 2), "valu_active_lanes_pct": round(pm.get("VALUUtilization", 0.0), 2),
                                           "valu_wave_instructions_per_launch": pm.get("SQ_INSTS_VALU"), "counters": note})
                                if pm.get("FETCH_SIZE") is not None and pm.get("WRITE_SIZE") is not None:
                                    tr = 2.0 * pm["FETCH_SIZE"] * 1024.0 + pm["WRITE_SIZE"] * 1024.0
                                    rf.update({"traffic": round(tr, 1), "hbm_measured_GBs": round(tr / ks_ / 1e9, 2), "hbm_measured_frac": round(tr / ks_ / 1e9 / HBM_PEAK_GBS, 5)})
                            else:
                                rf["counters"] = "not collected: %s" % note
                        row["roofline"] = rf
                    except Exception as e:
                        row["roofline_error"] = str(e)
                    return row
                extra["statue_depth1_1080p"] = rt_workload("statue.ply", 1920, 1080, 1)
                extra["dragon_4k"] = rt_workload("dragon_vis.ply", 3840, 2160, 3)
            # one 1024^2 shadow map (Light::CalculateXformFromWorldToLightSpace + RenderSceneIntoShadowBuffer, Light.cc:84-296): mi355_light_update
            # back to back on one stream, HIP events; k_sm_prep + k_sm_tiles (keys in LDS)
            sm = {}
            for mesh in ("chessboard.tri", args.mesh):
                sc = chess if mesh == "chessboard.tri" else scene
                for k in range(3):
                    sc.light_update(0, [3.394, 3.394, 4.8], 1024, stream.cuda_stream)
                torch.cuda.synchronize(dev)
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g0.record(stream)
                for k in range(50):
                    a = 0.785 + 0.01 * k
                    sc.light_update(0, [4.8 * float(np.cos(a)), 4.8 * float(np.sin(a)), 4.8], 1024, stream.cuda_stream)
                g1.record(stream)
                torch.cuda.synchronize(dev)
                sc.fetch_stats()
                sm[mesh] = round(g0.elapsed_time(g1) / 50 * 1e3, 1)
                sc.light_update(0, [float(v) for v in cams[0][1][0].pos], 1024, stream.cuda_stream)       # (the benchmark's light again)
                torch.cuda.synchronize(dev)
            extra["shadowmap_1024_us"] = sm
            # The reference's own benchmark procedure (`renderer -b -n N -m MODE FILE`, renderer.cc:243-341, 481-520) for BASELINE.json's five
            # configurations, through the C++ host layer's render_cli (scripts/render_cli_configs.sh); fewer passes than the script's
            cli = os.path.join(os.path.dirname(os.path.abspath(R.__file__)), "lib", "render_cli")
            if os.path.exists(cli) and not args.no_cli:
                import subprocess, re
                md = os.path.dirname(R.assets.mesh_path("chessboard.tri"))
                rows = []
                for label, a in (("1: chessboard.tri -m 2 640x480", ["-n", "1000", "-m", "2", "-W", "640", "-H", "480", "chessboard.tri"]),
                                 ("2: chessboard.tri -m 6 1920x1080", ["-n", "1000", "-m", "6", "-W", "1920", "-H", "1080", "chessboard.tri"]),
                                 ("3: statue.ply -m 9 --depth 1 1920x1080", ["-n", "500", "-m", "9", "--depth", "1", "-W", "1920", "-H", "1080", "statue.ply"]),
                                 ("4: dragon_vis.ply -m 9 1920x1080", ["-n", "500", "-m", "9", "-W", "1920", "-H", "1080", "dragon_vis.ply"]),
                                 ("5: dragon_vis.ply -m 9 3840x2160 (one GPU)", ["-n", "200", "-m", "9", "-W", "3840", "-H", "2160", "dragon_vis.ply"])):
                    row = {"config": label}
                    # (--keep-canvas: Screen::_keepCanvas, the front-end's promise that only Scene::render* writes into its canvases -- as
                    #  renderer.cc's loop does --: raster frames then cross PCIe only where they differ from the canvas's last frame)
                    kept = (("fps_in_flight_keep_canvas", ["--keep-canvas"]), ("fps_reference_loop_keep_canvas", ["-p", "1", "--keep-canvas"])) if "-m 2" not in label else ()
                    for key, pre in (("fps_3_in_flight", []), ("fps_reference_loop", ["-p", "1"])) + kept:
                        try:
                            out = subprocess.run([cli, "-b"] + pre + a[:-1] + [os.path.join(md, a[-1])], capture_output=True, text=True, timeout=60,
                                                 env=dict(os.environ, HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", str(local_rank))))
                            m = re.search(r"\(([0-9.]+) fps", out.stdout + out.stderr)
                            row[key] = float(m.group(1)) if m else None
                        except Exception as e:
                            row[key] = None
                            row["error"] = str(e)
                    rows.append(row)
                extra["render_cli_bench"] = {"rows": rows, "note": "render_cli -b: fps_3_in_flight = its default for -b (the cameras are known: three "
                                             "frames in flight through Scene::renderAsync); fps_reference_loop = -p 1, one synchronous Scene::render* per "
                                             "pass like renderer.cc:481-520, rate = frames / time inside the calls; *_keep_canvas (every row but the points; in flight: render_cli's default with --keep-canvas, four frames for the rasterizer, three for the raytracer): the "
                                             "same runs with --keep-canvas (mi355_opts::keep_canvas: the kernels write a frame straight into the page-locked "
                                             "canvas and only where it can differ from the canvas's last frame -- the rasterizer's 64x64-pixel bins that hold triangles now or held some then, the raytracer's 8x8-pixel tiles that are traced now or were then)"}
            # BVH build of the benchmark mesh: GPU level kernels + download + flatten, host builder beside it
            import ctypes as C
            bs = R.Scene(R.assets.mesh_path(args.mesh), device=local_rank)
            bs.context()
            bs.build_bvh_device()
            best = None
            for _ in range(6):
                t1 = time.perf_counter()
                bs.build_bvh_device()
                wall = (time.perf_counter() - t1) * 1e3          # the whole mi355_build_bvh call, tree installed
                tm = (C.c_double * 4)()
                R.lib().mi355i_bvh_last_times(tm)
                if best is None or wall < best[0]:
                    best = (wall, tm[1], tm[2], tm[3])
            extra["bvh_build_gpu_ms"] = round(best[0], 3)
            # (the build never leaves the device: level loop, pre-order numbering and the traversal streams are kernels)
            extra["bvh_build_gpu_parts_ms"] = {"all_kernels_one_sync": round(best[1], 3), "copy_tree_to_caller": round(best[2], 3),
                                               "install_in_context": round(best[3], 3)}
            t1 = time.perf_counter()
            bs.bvh_create("host")
            extra["bvh_build_host_cpu_ms"] = round((time.perf_counter() - t1) * 1e3, 1)
            # cold start: from the model file to the first raytraced 1080p frame in host memory (new context, GPU build)
            t1 = time.perf_counter()
            cs = R.Scene(R.assets.mesh_path(args.mesh), device=local_rank)
            t_load = time.perf_counter()
            cs.bvh_create()
            t_bvh = time.perf_counter()
            cs.render(9, cams[0][0], cams[0][1], cams[0][2], R.default_opts(W, H))
            extra["cold_start_ms"] = {"load_and_precompute": round((t_load - t1) * 1e3, 2), "upload_and_bvh_build": round((t_bvh - t_load) * 1e3, 2),
                                      "first_frame": round((time.perf_counter() - t_bvh) * 1e3, 2), "total": round((time.perf_counter() - t1) * 1e3, 2)}
        except Exception as e:      # secondary numbers must never break the headline line
            extra["error"] = str(e)
        result["other_workloads"] = extra

    # ---- CPU baseline: the oracle (a port of the reference's path) on this box's host cores, bounded sample
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            from oracle import oracle_ctypes as O
            ncpu = os.cpu_count() or 1
            osc = O.Scene(R.assets.mesh_path(args.mesh))
            if args.mode >= 9:
                osc.bvh_ensure(os.path.join(R.assets.cache_dir(), args.mesh + ".oracle.bvh"))
            # the port scales to a box-dependent thread count (OpenMP over scanlines, like the reference): probe
            # a few counts on frame 0 and time the sample with the fastest one
            maps = None
            ocam, olights, on = O.benchmark_frame(0)
            if args.mode in (7, 8):
                maps = [osc.shadowmap(olights[0])]
            best_t, best_dt = 1, None
            for t in sorted({1, 8, 16, 32, 64, ncpu}):
                if t > ncpu:
                    continue
                ot = O.default_opts(W, H, threads=t)
                osc.render(args.mode, ocam, olights, on, ot, shadow_maps=maps)
                t1 = time.perf_counter()
                osc.render(args.mode, ocam, olights, on, ot, shadow_maps=maps)
                d = time.perf_counter() - t1
                if best_dt is None or d < best_dt:
                    best_t, best_dt = t, d
            oo = O.default_opts(W, H, threads=best_t)
            done_rays, done_frames, t_cpu = 0.0, 0, 0.0
            k = 0
            while (t_cpu < args.cpu_seconds and k < max(K, 100)) or k < 100 or t_cpu < 3.0:
                ocam, olights, on = O.benchmark_frame(k)
                t1 = time.perf_counter()
                _, _, st = osc.render(args.mode, ocam, olights, on, oo, shadow_maps=maps)
                t_cpu += time.perf_counter() - t1
                done_rays += st.normal_rays + st.shadow_rays
                done_frames += 1
                k += 1
            port = {
                "value": round(done_rays / t_cpu / 1e6, 3), "unit": "Mrays/s", "cores": cpu_grant()["granted_cores"], "threads": best_t, "kind": "port",
                "sample": "oracle (strict-IEEE C++ port of the reference, OpenMP over pixels like Raytracer.cc:558) on "
                          "frames f0..f%d of the same workload, %.1f s; %d threads = fastest of a probe over {1,8,16,32,64,%d} on this %d-thread host"
                          % (done_frames - 1, t_cpu, best_t, ncpu, ncpu),
                "frames_per_sec": round(done_frames / t_cpu, 3),
            }
            result["cpu_baseline"] = port
            # ---- the REFERENCE's own code as the baseline (north_star: "the reference's OpenMP/SSE CPU path is timed on the same
            #      box"): oracle/_ref/refcore_omp = Raytracer.cc itself (RayIntersectsBox, BVH_IntersectTriangles<>, Raytrace<true>),
            #      compiled from /root/reference with the pinned strict flags, under refcore.cc's frame loop (OpenMP over pixels; both
            #      the reference's per-scanline shape and a per-frame loop are probed, the faster one is timed).  Rays per frame are
            #      the counting build's (= the reference's own counters) for the same cameras.
            if args.mode == 9:
                try:
                    from oracle import refcore as RC
                    if RC.timing_available():
                        ocams = [O.benchmark_frame(f) for f in range(N_CAMS)]
                        lights0, n0 = ocams[0][1], ocams[0][2]
                        # (three frames per probe, the first dropped: the box's CPU quota lets a short burst run faster than anything
                        #  sustained, and a parallel-for per scanline on hundreds of threads can take minutes -- both bounded here)
                        probe = {}
                        for sched, counts in ((1, {16, 32, 64, 128}), (0, {16, 32})):
                            for t in sorted(c for c in counts if c <= ncpu):
                                try:
                                    secs, _ = RC.time_frames(osc, [ocams[k][0] for k in (0, 1, 2)], lights0, n0, W, H, 2 * H, threads=t, schedule=sched, timeout=40)
                                    probe[(sched, t)] = float(secs[1:].mean())
                                except Exception:
                                    probe[(sched, t)] = float("inf")
                        (b_sched, b_t), b_s = min(probe.items(), key=lambda kv: kv[1])
                        sample = [f for f in used if rays_f[f] > 0][:max(4, min(len(used), int(round(args.cpu_seconds / max(b_s, 1e-3)))))]
                        # (the sample twice: the box's CPU quota makes single samples move by a quarter from box to box; the better run counts)
                        grant = cpu_grant()
                        snaps = [cpu_stat()]
                        runs = []
                        for _ in range(2):
                            runs.append(RC.time_frames(osc, [ocams[f][0] for f in sample], lights0, n0, W, H, 2 * H, threads=b_t, schedule=b_sched)[0])
                            snaps.append(cpu_stat())
                        best_run = min(range(2), key=lambda i: float(runs[i].sum()))
                        secs = runs[best_run]
                        usage = cpu_stat_delta(snaps[best_run], snaps[best_run + 1], b_t)
                        r_rays, r_t = float(sum(rays_f[f] for f in sample)), float(secs.sum())
                        ref_single = RC.time_frames(osc, [ocams[0][0]], lights0, n0, W, H, 2 * H, threads=1, schedule=1)[0][0] if args.cpu_seconds >= 8 else None
                        # `cores` = what the box grants this process (its CFS quota if it has one, else its affinity mask), `threads` = what
                        # was launched; `burst_Mrays_per_s` = the fastest probe (two frames: what the host does before any quota bites)
                        burst = float(np.mean([rays_f[1], rays_f[2]])) / b_s / 1e6 if b_s > 0 and rays_f[1] > 0 and rays_f[2] > 0 else None
                        thr = usage.get("throttled_fraction")
                        note = ("; the cgroup's CFS quota throttled this sample in %.0f %% of its periods" % (100 * thr)) if (thr is not None and thr > 0.10) else ""
                        if usage["effective_cores_used"] < 0.5 * min(b_t, grant["granted_cores"]):
                            note += ("; the sample got %.1f cores' worth of CPU time for its %d threads (the host does not grant more, whatever it shows)"
                                     % (usage["effective_cores_used"], b_t))
                        result["cpu_baseline"] = {
                            "value": round(r_rays / r_t / 1e6, 3), "unit": "Mrays/s", "cores": grant["granted_cores"], "threads": b_t, "kind": "reference",
                            "cpu_quota_cores": grant["cpu_quota_cores"], "host_threads": grant["host_threads"], "affinity_threads": grant["affinity_threads"],
                            "effective_cores_used": usage["effective_cores_used"], "throttled_fraction": thr,
                            "throttled_seconds_per_wall_second": usage.get("throttled_seconds_per_wall_second"),
                            "burst_Mrays_per_s": None if burst is None else round(burst, 2),
                            "sample": "oracle/_ref/refcore_omp: the reference's own Raytracer.cc (Raytrace<true>, strict flags -O2 -ffp-contract=off) "
                                      "on orbit frames %s of the same workload (%d frames, %.1f s), OpenMP over pixels in refcore.cc's frame loop (%s), "
                                      "%d threads = fastest of a probe over threads x loop shape on this %d-thread host; rays per frame from the "
                                      "counting build (= the reference's counters)%s"
                                      % ("f%d..f%d" % (sample[0], sample[-1]), len(sample), r_t,
                                         "one parallel loop over the frame's scanlines" if b_sched == 1 else "the reference's shape: a parallel-for over x per scanline",
                                         b_t, ncpu, note),
                            "frames_per_sec": round(len(sample) / r_t, 3),
                            "probe_ms_per_frame": {"%s/%dt" % ("rows" if k[0] == 1 else "per-scanline", k[1]): (round(v * 1e3, 1) if v < 1e9 else "timed out")
                                                   for k, v in sorted(probe.items())},
                        }
                        result["cpu_baseline"]["runs_Mrays_per_s"] = [round(r_rays / float(v.sum()) / 1e6, 2) for v in runs]
                        if ref_single:
                            result["cpu_baseline"]["single_thread_Mrays_per_s"] = round(float(rays_f[0]) / float(ref_single) / 1e6, 3)
                        result["cpu_baseline_port"] = port
                        # ... and the reference's SHIPPING configuration: the same code and frame loop with the flags its own configure.ac
                        # gives a release build (-O3 -ffast-math -funsafe-math-optimizations -mrecip -msse2 ... -DSIMD_SSE): faster, and
                        # not bit-compatible with the strict build the parity is pinned to (92 pixels of a dragon frame move)
                        if os.path.exists(RC.AUTHOR_BINARY):
                            aruns = [RC.time_frames(osc, [ocams[f][0] for f in sample], lights0, n0, W, H, 2 * H, threads=b_t, schedule=b_sched, binary=RC.AUTHOR_BINARY)[0] for _ in range(2)]
                            a_t = min(float(v.sum()) for v in aruns)
                            result["cpu_baseline_author"] = {
                                "value": round(r_rays / a_t / 1e6, 3), "unit": "Mrays/s", "cores": grant["granted_cores"], "threads": b_t, "kind": "reference",
                                "sample": "oracle/_ref/refcore_omp_author: the same Raytracer.cc and frame loop with the reference's own release flags "
                                          "(configure.ac:47-50, 193-264: -O3 -fomit-frame-pointer -ffast-math -funsafe-math-optimizations -mtune=native -flto "
                                          "-msse -mrecip -mfpmath=sse -msse2 -mssse3 -DSIMD_SSE -DSIMD_SSE2 -DNDEBUG), the same %d frames, threads and loop "
                                          "shape as cpu_baseline; the better of two runs" % len(sample),
                                "frames_per_sec": round(len(sample) / a_t, 3), "runs_Mrays_per_s": [round(r_rays / float(v.sum()) / 1e6, 2) for v in aruns],
                                "x_strict": round((r_rays / a_t) / (r_rays / r_t), 3)}
                except Exception as e:
                    result["cpu_baseline_reference_error"] = str(e)
            # the rasterizer's CPU baseline: the oracle draws triangles in index order on ONE thread -- the only deterministic
            # semantics the reference has (its OpenMP build races on the Z-buffer)
            if not args.no_extra:
                oc = O.Scene(R.assets.mesh_path("chessboard.tri"))
                # BASELINE configs[0]: chessboard.tri, points-only rasterizer (mode 2), 640x480, CPU path, single thread
                n_c, t_c = 0, 0.0
                while t_c < 2.0 or n_c < 200:
                    ocam, olights, on = O.benchmark_frame(n_c % N_CAMS)
                    t1 = time.perf_counter()
                    oc.render(2, ocam, olights, on, O.default_opts(640, 480, threads=1))
                    t_c += time.perf_counter() - t1
                    n_c += 1
                result["cpu_baseline_config1"] = {"value": round(n_c / t_c, 1), "unit": "frames/s", "cores": 1, "kind": "port",
                                                  "sample": "oracle, chessboard.tri, mode 2 (points from triangles), 640x480, single thread, %d orbit frames, %.1f s "
                                                            "(BASELINE configs[0]; the GPU's rate for the same frames: other_workloads.render_cli_bench row 1)" % (n_c, t_c)}
                for mode, name in ((6, "chessboard_phong_1080p"), (8, "chessboard_softshadow_1080p")):
                    ocam, olights, on = O.benchmark_frame(0)
                    rmaps = [oc.shadowmap(olights[0])] if mode == 8 else None
                    n_c, t_c = 0, 0.0
                    while t_c < 3.0 or n_c < 20:
                        ocam, olights, on = O.benchmark_frame(n_c)
                        t1 = time.perf_counter()
                        oc.render(mode, ocam, olights, on, O.default_opts(W, H, threads=1), shadow_maps=rmaps)
                        t_c += time.perf_counter() - t1
                        n_c += 1
                    result.setdefault("cpu_baseline_raster", {})[name] = {
                        "value": round(n_c / t_c, 2), "unit": "frames/s", "cores": 1, "kind": "port",
                        "sample": "oracle, single thread (triangles in index order), frames f0..f%d, %.1f s" % (n_c - 1, t_c)}
        except Exception as e:
            result["cpu_baseline"] = {"value": None, "unit": "Mrays/s", "cores": 0, "kind": "port", "sample": "failed: %s" % e}

    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
