#!/usr/bin/env python3
"""Distil the rocprofv3 output of scripts/gpu_round_full.sh (under gpurun_out/) into the tracked files in profiles/.

    python scripts/make_profiles.py [round-tag, default r06]
"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TAG = sys.argv[1] if len(sys.argv) > 1 else "r06"
KERNELS = {"default": "k_raytrace<false, false, true, 4, true", "noshare": "k_raytrace<false, false, true, 4, true", "bpc3": "k_raytrace<false, false, true, 3, true"}


def newest(pattern):
    files = sorted(glob.glob(os.path.join(ROOT, pattern), recursive=True), key=os.path.getmtime)
    return files[-1] if files else None


def per_launch(variant):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for d in glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_%s_*" % variant)):
        if not os.path.isdir(d):
            continue
        # (gpurun_out/ is merged from call to call: only the newest pass of each directory counts)
        for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]:
            for row in csv.DictReader(open(f)):
                if KERNELS[variant] in row["Kernel_Name"]:
                    a = acc[row["Counter_Name"]]
                    a[0] += float(row["Counter_Value"]); a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}, {k: v[1] for k, v in acc.items()}


def main():
    out = os.path.join(ROOT, "profiles")
    for src, dst in (("prof_stats", "kernel_stats"), ("prof_stats_overlapped", "kernel_stats_overlapped"), ("prof_stats_raster", "kernel_stats_raster"),
                     ("prof_stats_shadowmap", "kernel_stats_shadowmap")):
        f = newest("gpurun_out/%s/**/*kernel_stats.csv" % src)
        if f:
            shutil.copy(f, os.path.join(out, "%s_%s.csv" % (TAG, dst)))
    variants = {}
    for v in KERNELS:
        pmc, launches = per_launch(v)
        if not pmc:
            continue
        fetch_kb, write_kb = pmc.get("FETCH_SIZE"), pmc.get("WRITE_SIZE")
        variants[v] = {"kernel": KERNELS[v] + ", ...>", "pmc_per_launch": pmc, "launches": launches,
                       "hbm_bytes_per_launch": (2.0 * fetch_kb * 1024.0 + write_kb * 1024.0) if fetch_kb is not None and write_kb is not None else None}
    if variants:
        old = {}
        tf = os.path.join(out, "traffic.json")
        if os.path.exists(tf):
            try: old = json.load(open(tf))
            except Exception: old = {}
        d = variants.get("default", {})
        json.dump({
            "round": TAG,
            "workload": "dragon_vis.ply mode 9 1920x1080, 8 frames per launch: bench.py --pmc-child under rocprofv3 --pmc (one pass per counter group; launches one after the other)",
            "fetch_correction": "x2 (gfx950 rocprofv3 tallies 128-B requests at 64 B for 16 B/lane loads; MI355X_MICROARCH.md HBM section). WRITE_SIZE uncorrected.",
            "kernel": d.get("kernel"), "pmc_per_launch": d.get("pmc_per_launch"), "launches": d.get("launches"), "k_raytrace_hbm_bytes_per_launch": d.get("hbm_bytes_per_launch"),
            "variants": {"default": "four waves per SIMD, work shared inside a wave", "noshare": "the same build, tune flag 256 (no sharing)", "bpc3": "three waves per SIMD (no scratch), work shared"},
            "by_variant": variants,
            "raster_kernels": old.get("raster_kernels"), "raster_kernels_round": old.get("round") if old.get("raster_kernels") else None,
        }, open(tf, "w"), indent=1)
    for src, dst in (("gpurun_out/bench_full.log", "%s_bench_n1.jsonl" % TAG), ("gpurun_out/bench_dryrun_n2.log", "%s_bench_dryrun_n2.jsonl" % TAG),
                     ("gpurun_out/bench_dryrun_n8.log", "%s_bench_dryrun_n8.jsonl" % TAG), ("gpurun_out/pytest_full.log", "%s_pytest_gpu.log" % TAG),
                     ("gpurun_out/misc_full.log", "%s_side_measurements.log" % TAG), ("gpurun_out/rt_pmc.json", "%s_pmc_bench_kernel.json" % TAG)):
        p = os.path.join(ROOT, src)
        if os.path.exists(p):
            lines = [l for l in open(p).read().splitlines() if l.strip()]
            open(os.path.join(out, dst), "w").write("\n".join(lines[-1:] if dst.endswith("jsonl") else lines) + "\n")
    for v, d in variants.items():
        p = d["pmc_per_launch"]
        print(v, {k: round(p[k], 2) for k in ("VALUBusy", "VALUUtilization", "SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE") if k in p}, "HBM bytes/launch", d["hbm_bytes_per_launch"])
    ks = newest("gpurun_out/prof_stats/**/*kernel_stats.csv")
    if ks:
        for i, row in enumerate(csv.DictReader(open(ks))):
            if i < 6:
                print(row["Name"][:80], row["Calls"], row["AverageNs"])


if __name__ == "__main__":
    main()
