#!/usr/bin/env python3
"""Kernel variants of k_raytrace side by side on the bench workload.

    python scripts/rt_variants.py [name[:ENV=V[;ENV=V]] ...]      (name "default" = the committed library)

Every variant is renderer_amd/lib/variant_<name>.so (scripts/build_rt_variant.sh) run in a process of its own (MI355_RENDER_SO);
per variant one JSON line: frames/s of dragon 1080p depth 3 in batches of 8 (overlapped launches, as bench.py times them), the
same for statue depth 1, kernel ms of single frames, and SHA-256 of orbit frames 0 / 37 / 100 -- frame 0 must be the reference's
pin (tests/golden/reference_pins.json), the others must agree across variants."""
import hashlib, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def child():
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    import renderer_amd as R
    dev = torch.device("cuda", 0)
    W, H, B = 1920, 1080, int(os.environ.get('RT_B', '8'))
    stream = torch.cuda.current_stream(dev)
    cams = [R.benchmark_frame(k) for k in range(200)]
    bufs = [torch.zeros((H, W), dtype=torch.int32, device=dev) for _ in range(B)]
    out = {"variant": os.environ.get("RT_VARIANT_NAME", "?")}
    for mesh, depth, tag in (("dragon_vis.ply", 3, "dragon"), ("statue.ply", 1, "statue")):
        s = R.Scene(R.assets.mesh_path(mesh)); s.bvh_create()
        o = R.default_opts(W, H, max_ray_depth=depth, tune=R.tune(**json.loads(os.environ.get('RT_TUNE', '{}'))))
        def step(i):
            ks = [(i * B + j) % 200 for j in range(B)]
            s.render_batch_device(9, [cams[k][0] for k in ks], [cams[k][1] for k in ks], 1, o, [b.data_ptr() for b in bufs], W * 4, None, stream.cuda_stream)
        for i in range(6): step(i)
        torch.cuda.synchronize(dev)
        rates = []
        for rep in range(4):
            t0 = time.perf_counter()
            for i in range(400 // B): step(i)
            torch.cuda.synchronize(dev)
            rates.append((400 // B) * B / (time.perf_counter() - t0))
        out[tag + "_batch8_fps"] = round(max(rates), 1)
        out[tag + "_batch8_fps_all"] = [round(r) for r in rates]
        ms = []
        for k in list(range(0, 200, 10)) * 2:
            cam, lights, n = cams[k]
            _, _, st = s.render(9, cam, lights, n, o)
            ms.append(st.kernel_ms)
        out[tag + "_single_ms"] = round(float(np.mean(ms[20:])), 4)
        hs = []
        for k in (0, 37, 100):
            px, _, _ = s.render(9, cams[k][0], cams[k][1], cams[k][2], o)
            a = np.asarray(px, dtype=np.uint32)
            rgb = np.stack([(a >> 16) & 255, (a >> 8) & 255, a & 255], axis=-1).astype(np.uint8)
            hs.append(hashlib.sha256(rgb.tobytes()).hexdigest()[:16])
        out[tag + "_sha"] = hs
    print(json.dumps(out), flush=True)

def main():
    names = sys.argv[1:] or ["default"]
    first = None
    for spec in names:
        name, _, envs = spec.partition(":")
        env = dict(os.environ, RT_VARIANT_NAME=spec, RT_VARIANT_CHILD="1")
        if name != "default": env["MI355_RENDER_SO"] = os.path.join(ROOT, "renderer_amd", "lib", "variant_%s.so" % name)
        for kv in filter(None, envs.split(";")):
            k, _, v = kv.partition("="); env[k] = v
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=240)
        except subprocess.TimeoutExpired:
            print(json.dumps({"variant": spec, "error": "timeout"}), flush=True); continue
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line: print(json.dumps({"variant": spec, "error": (r.stderr or r.stdout)[-600:]}), flush=True); continue
        d = json.loads(line[-1])
        if first is None: first = d
        d["same_as_first"] = all(d.get(k) == first.get(k) for k in ("dragon_sha", "statue_sha"))
        print(json.dumps(d), flush=True)

if __name__ == "__main__":
    child() if os.environ.get("RT_VARIANT_CHILD") else main()
