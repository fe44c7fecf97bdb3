#!/usr/bin/env python3
"""Variants of the rasterizer's kernels side by side on the chessboard at 1920x1080.

    python scripts/rs_variants.py [name[:ENV=V[;ENV=V]] ...]      (name "default" = the committed library)

Every variant is renderer_amd/lib/variant_<name>.so (scripts/build_rs_variant.sh; variant_base.so = a copy of an earlier library) run
in a process of its own (MI355_RENDER_SO); RS_TUNE = JSON arguments of renderer_amd.tune().  Per variant one JSON line: frames/s
frame by frame along the orbit (2 000 frames, best of three, as bench.py times the raster rows) for modes 4 / 6 / 8, batches of 8,
kernel ms of synchronous single frames, the 1024^2 shadow map's time, and SHA-256 of frames 0 / 37 / 100 of modes 6 and 8 -- they
must agree across variants (and frame 0 is the reference's pin, which the GPU suite checks)."""
import hashlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    import renderer_amd as R
    dev = torch.device("cuda", 0)
    W, H, B = 1920, 1080, 8
    stream = torch.cuda.current_stream(dev)
    cams = [R.benchmark_frame(k) for k in range(200)]
    s = R.Scene(R.assets.mesh_path(os.environ.get("RS_MESH", "chessboard.tri")))
    s.shadowmap_render(0, cams[0][1][0])
    buf = torch.zeros((H, W), dtype=torch.int32, device=dev)
    bufs = [torch.zeros((H, W), dtype=torch.int32, device=dev) for _ in range(B)]
    o = R.default_opts(W, H, tune=R.tune(**json.loads(os.environ.get("RS_TUNE", "{}"))))
    out = {"variant": os.environ.get("RS_VARIANT_NAME", "?")}
    n = int(os.environ.get("RS_FRAMES", "2000"))
    for mode in (6, 8, 4):
        for k in range(20): s.render_device(mode, *cams[k], o, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
        torch.cuda.synchronize(dev)
        rates = []
        for rep in range(3):
            t = time.perf_counter()
            for k in range(n): s.render_device(mode, *cams[k % 200], o, buf.data_ptr(), W * 4, 0, stream.cuda_stream)
            torch.cuda.synchronize(dev); rates.append(n / (time.perf_counter() - t))
        out["mode%d_fps" % mode] = round(max(rates)); out["mode%d_fps_all" % mode] = [round(r) for r in rates]
    for mode in (6, 8):
        def step(i):
            ks = [(i * B + j) % 200 for j in range(B)]
            s.render_batch_device(mode, [cams[k][0] for k in ks], [cams[k][1] for k in ks], 1, o, [b.data_ptr() for b in bufs], W * 4, None, stream.cuda_stream)
        for i in range(4): step(i)
        torch.cuda.synchronize(dev)
        rates = []
        for rep in range(3):
            t = time.perf_counter()
            for i in range(100): step(i)
            torch.cuda.synchronize(dev); rates.append(100 * B / (time.perf_counter() - t))
        out["mode%d_batch8_fps" % mode] = round(max(rates))
        ms = []
        for k in list(range(0, 200, 10)) * 2:
            _, _, st = s.render(mode, *cams[k], o)
            ms.append(st.kernel_ms)
        out["mode%d_single_ms" % mode] = round(float(np.mean(ms[20:])), 4)
        hs = []
        for k in (0, 37, 100):
            a = np.asarray(s.render(mode, *cams[k], o)[0], dtype=np.uint32)
            rgb = np.stack([(a >> 16) & 255, (a >> 8) & 255, a & 255], axis=-1).astype(np.uint8)
            hs.append(hashlib.sha256(rgb.tobytes()).hexdigest()[:16])
        out["mode%d_sha" % mode] = hs
    # the 1024^2 shadow map, back to back on one stream (scripts/shadowmap_time.py), for three meshes; the chessboard's map hashed
    for mesh in ("chessboard.tri", "dragon_vis.ply", "statue.ply"):
        sm = s if mesh == os.environ.get("RS_MESH", "chessboard.tri") else R.Scene(R.assets.mesh_path(mesh))
        for _ in range(5): sm.light_update(0, [3.394, 3.394, 4.8], 1024, stream.cuda_stream)
        torch.cuda.synchronize(dev)
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for k in range(100):
                a = 0.785 + 0.01 * k
                sm.light_update(0, [4.8 * np.cos(a), 4.8 * np.sin(a), 4.8], 1024, stream.cuda_stream)
            e1.record(stream); torch.cuda.synchronize(dev)
            best = min(best, e0.elapsed_time(e1) / 100 * 1e3)
        out["shadowmap_us_" + mesh.split(".")[0].split("_")[0]] = round(best, 1)
        if sm is s:
            out["shadowmap_sha"] = hashlib.sha256(np.asarray(s.shadowmap_render(0, cams[0][1][0], 1024, fetch=True)).tobytes()).hexdigest()[:16]
    print(json.dumps(out), flush=True)


def main():
    names = sys.argv[1:] or ["default"]
    first = None
    for spec in names:
        name, _, envs = spec.partition(":")
        env = dict(os.environ, RS_VARIANT_NAME=spec, RS_VARIANT_CHILD="1")
        if name != "default": env["MI355_RENDER_SO"] = os.path.join(ROOT, "renderer_amd", "lib", "variant_%s.so" % name)
        for kv in filter(None, envs.split(";")):
            k, _, v = kv.partition("="); env[k] = v
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=300)
        except subprocess.TimeoutExpired:
            print(json.dumps({"variant": spec, "error": "timeout"}), flush=True); continue
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line: print(json.dumps({"variant": spec, "error": (r.stderr or r.stdout)[-600:]}), flush=True); continue
        d = json.loads(line[-1])
        if first is None: first = d
        d["same_as_first"] = all(d.get(k) == first.get(k) for k in ("mode6_sha", "mode8_sha", "shadowmap_sha"))
        print(json.dumps(d), flush=True)


if __name__ == "__main__":
    child() if os.environ.get("RS_VARIANT_CHILD") else main()
