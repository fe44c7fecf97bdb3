#!/usr/bin/env python3
"""Distil the rocprofv3 output of scripts/gpu_round2_full.sh (under gpurun_out/) into the tracked files in profiles/.

    python scripts/make_profiles.py [round-tag, default r02]
"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r02"
KERNEL = "k_raytrace<false, false, true, 4, true, false>"   # the bench kernel: ordered walk, batched launch, four waves per SIMD


def newest(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, pattern), recursive=True), key=os.path.getmtime)
    return files[-1] if files else None


def per_launch(kind, kernel=KERNEL):
    f = newest("gpurun_out/%s/**/*counter_collection.csv" % kind)
    acc = collections.defaultdict(lambda: [0.0, 0])
    if f:
        for row in csv.DictReader(open(f)):
            if kernel in row["Kernel_Name"]:
                a = acc[row["Counter_Name"]]
                a[0] += float(row["Counter_Value"]); a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}, {k: v[1] for k, v in acc.items()}


def main():
    out = os.path.join(ROOT, "profiles")
    ks = newest("gpurun_out/prof_stats/**/*kernel_stats.csv")
    if ks:
        shutil.copy(ks, os.path.join(out, "%s_kernel_stats.csv" % TAG))
    for kind in ("raster", "raster_overlapped"):      # (scripts/gpu_round2_final.sh: the rasterizer's kernels, mode 6, one frame per call)
        kr = newest("gpurun_out/prof_stats_%s/**/*kernel_stats.csv" % kind)
        if kr:
            shutil.copy(kr, os.path.join(out, "%s_kernel_stats_%s.csv" % (TAG, kind)))
    ko = newest("gpurun_out/prof_stats_overlapped/**/*kernel_stats.csv")
    if ko:
        shutil.copy(ko, os.path.join(out, "%s_kernel_stats_overlapped.csv" % TAG))
    pmc, launches = {}, {}
    for kind in ("prof_fetch", "prof_write", "prof_sq", "prof_cache", "prof_valu1", "prof_valu2"):
        v, n = per_launch(kind)
        pmc.update(v); launches.update(n)
    fetch_kb, write_kb = pmc.get("FETCH_SIZE"), pmc.get("WRITE_SIZE")
    traffic = None
    if fetch_kb is not None and write_kb is not None:
        traffic = 2.0 * fetch_kb * 1024.0 + write_kb * 1024.0
    # the rasterizer's three tiled kernels (mode 6, chessboard 1080p, single-frame launches)
    raster = {}
    for kern in ("k_rs_setup", "k_rs_fill", "k_rs_tile"):
        r, rn = {}, {}
        for kind in ("prof_rs_fetch", "prof_rs_write", "prof_rs_sq", "prof_rs_valu"):
            v, n = per_launch(kind, kern)
            r.update(v); rn.update(n)
        if r:
            f_kb, w_kb = r.get("FETCH_SIZE"), r.get("WRITE_SIZE")
            raster[kern] = {"pmc_per_launch": r, "launches": max(rn.values()),
                            "hbm_bytes_per_launch": (2.0 * f_kb * 1024.0 + w_kb * 1024.0) if f_kb is not None and w_kb is not None else None}
    json.dump({
        "round": TAG,
        "raster_kernels": raster,
        "kernel": KERNEL,
        "workload": "dragon_vis.ply mode 9 1920x1080, bench.py --steps 20 --warmup 2 under rocprofv3 --pmc (one pass per counter group)",
        "FETCH_SIZE_KB_per_launch_raw": fetch_kb, "WRITE_SIZE_KB_per_launch_raw": write_kb,
        "fetch_correction": "x2 (gfx950 rocprofv3 tallies 128-B requests at 64 B for 16 B/lane loads; MI355X_MICROARCH.md HBM section). WRITE_SIZE uncorrected.",
        "k_raytrace_hbm_bytes_per_launch": traffic,
        "pmc_per_launch": pmc, "launches": launches,
    }, open(os.path.join(out, "traffic.json"), "w"), indent=1)
    for src, dst in (("gpurun_out/bench_full.log", "%s_bench_n1.jsonl" % TAG), ("gpurun_out/pytest_full.log", "%s_pytest_gpu.log" % TAG),
                     ("gpurun_out/misc_full.log", "%s_side_measurements.log" % TAG)):
        p = os.path.join(ROOT, src)
        if os.path.exists(p):
            lines = [l for l in open(p).read().splitlines() if l.strip()]
            open(os.path.join(out, dst), "w").write("\n".join(lines[-1:] if dst.endswith("jsonl") else lines) + "\n")
    mo = os.path.join(ROOT, "gpurun_out/misc_overlap.log")
    if os.path.exists(mo):          # (scripts/gpu_round2_final.sh: the frame-by-frame measurements after the overlap work)
        with open(os.path.join(out, "%s_side_measurements.log" % TAG), "a") as f:
            f.write("".join(l for l in open(mo) if l.strip()))
            for extra, title in (("gpurun_out/rt_fbf_bpc.txt", "scripts/raytrace_frame_by_frame.py: waves per SIMD of overlapped single frames (tune bpc)"),
                                 ("gpurun_out/rt_fbf_final.txt", "scripts/raytrace_frame_by_frame.py after overlapped single frames took the four-wave build"),
                                 ("gpurun_out/wire_fps.txt", "wireframe (mode 3) chessboard 1080p after the thread-per-operation rewrite (1229.9 fps before)")):
                q = os.path.join(ROOT, extra)
                if os.path.exists(q):
                    f.write("== %s\n" % title + "".join(l for l in open(q) if l.strip()))
    print("traffic bytes/launch:", traffic, "| launches:", launches)
    if ks:
        for i, row in enumerate(csv.DictReader(open(ks))):
            if i < 6:
                print(row["Name"][:80], row["Calls"], row["AverageNs"])


if __name__ == "__main__":
    main()
