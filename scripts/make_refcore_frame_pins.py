#!/usr/bin/env python3
"""tests/golden/refcore_frame_pins.json: SHA-256 of full-size raytraced orbit frames BEYOND the first one, from the reference's own
code run here.  (tests/golden/reference_pins.json holds frame f0 of every BASELINE configuration, recorded by the survey from a
whole build of the program; nobody can regenerate those in this image.)

For BASELINE configs[2] (statue.ply, max depth 1) and configs[3] (dragon_vis.ply, depth 3) at 1920x1080, orbit frames f37, f100, f150,
configs[4]'s 3840x2160 at f37 and f100, the 4 spp mode (10) and frames with the second light:
every pixel's colour is Raytrace<true>() of /root/reference/src/Raytracer.cc, compiled from where it lies with the pinned strict flags
(oracle/refcore/Makefile -> oracle/_ref/refcore, command `raytrace`), on the frame's camera rays; the packed pixel is that colour
clamped at 255 and truncated as Raytracer.cc:600-604 does.  Frame f0 made the same way must reproduce the survey's pin -- the check
that this recipe IS the reference's frame -- and the script refuses to write anything if it does not.

Needs /root/reference (to build oracle/_ref) -- i.e. this container, not the GPU box."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import oracle_ctypes as O, refcore as RC
import renderer_amd.assets as A

def frame(osc, k, w, h, depth, mode=9, two=False):
    cam, lights, n = O.benchmark_frame(k, two)
    if mode == 9:
        f = RC.raytrace(osc, cam, lights, n, RC.primary_rays(cam, w, h, 2 * h), depth).reshape(h, w, 3)
    else:                                                           # mode 10: while(pixelsTraced--), Raytracer.cc:570-597, then / 4
        f = np.zeros((h, w, 3), np.float32)
        for sub in (3, 2, 1, 0):
            f = f + RC.raytrace(osc, cam, lights, n, RC.primary_rays(cam, w, h, 2 * h, sub=sub), depth).reshape(h, w, 3)
        f = f / np.float32(4.0)
    c = np.minimum(f, np.float32(255.0)).astype(np.uint32)          # Raytracer.cc:600-603, then (Uint8) truncation
    rgb = np.stack([c[..., 0], c[..., 1], c[..., 2]], axis=-1).astype(np.uint8)
    return f, rgb

def winners_hash(tri, passes, fat):
    bits = np.ascontiguousarray(fat, np.float32).view(np.uint32).copy()
    bits[np.isnan(fat)] = 0x7fc00000
    bits[tri < 0] = 0
    h = hashlib.sha256()
    for x in (np.ascontiguousarray(tri, np.int32), np.ascontiguousarray(passes, np.int32), bits):
        h.update(x.tobytes())
    return h.hexdigest()

def main():
    assert RC.build(), "oracle/_ref/refcore could not be built (is /root/reference present?)"
    survey = {p["id"]: p for p in json.load(open(os.path.join(ROOT, "tests", "golden", "reference_pins.json")))["frames"]}
    out = []
    scenes = {}
    def scene(mesh):
        if mesh not in scenes:
            scenes[mesh] = O.Scene(A.mesh_path(mesh)); scenes[mesh].bvh_build()
        return scenes[mesh]
    for cid, mesh, depth in (("cfg3", "statue.ply", 1), ("cfg4", "dragon_vis.ply", 3)):
        _, rgb0 = frame(scene(mesh), 0, 1920, 1080, depth)
        assert hashlib.sha256(rgb0.tobytes()).hexdigest() == survey[cid]["sha256"], "%s: frame f0 made this way is not the survey's frame" % cid
    # (id, mesh, mode, w, h, depth, frames, second light): the two raytrace configurations along the orbit; config 5's size; the 4 spp
    # mode; two lights
    for cid, mesh, mode, w, h, depth, frames, two in (
            ("cfg3", "statue.ply", 9, 1920, 1080, 1, (37, 100, 150), False), ("cfg4", "dragon_vis.ply", 9, 1920, 1080, 3, (37, 100, 150), False),
            ("cfg5", "dragon_vis.ply", 9, 3840, 2160, 3, (37, 100), False), ("cfg4_aa", "dragon_vis.ply", 10, 1920, 1080, 3, (37,), False),
            ("cfg3_aa_2lights", "statue.ply", 10, 1920, 1080, 1, (100,), True), ("cfg4_2lights", "dragon_vis.ply", 9, 1920, 1080, 3, (150,), True)):
        for k in frames:
            f, rgb = frame(scene(mesh), k, w, h, depth, mode, two)
            out.append({"id": "%s_f%d" % (cid, k), "mesh": mesh, "mode": mode, "w": w, "h": h, "depth": depth, "frame": k, "second_light": two,
                        "nonblack": int((rgb.astype(np.uint32).sum(-1) != 0).sum()), "sha256": hashlib.sha256(rgb.tobytes()).hexdigest(),
                        "sha256_f32": hashlib.sha256(np.minimum(f, np.float32(255.0)).astype(np.float32).tobytes()).hexdigest()})
            print(out[-1], flush=True)
    # the rasterizer (BASELINE configs[1]: chessboard.tri, mode 6) along the orbit: what the reference's own DrawTriangles / Filler<> /
    # ScanConverter / RasterizeTriangle hand its plotter (oracle/_ref/refraster, at the reference's compile-time 800 x 600)
    rout = []
    if RC.raster_available():
        osc = O.Scene(A.mesh_path("chessboard.tri"))
        for k in (0, 37, 100, 150):
            cam, lights, n = O.benchmark_frame(k)
            W, H, mv, tri, passes, fat = RC.raster_winners(osc, 6, list(cam.eye), [0.0, 0.0, 0.0], [list(lights[0].pos)])
            assert np.array_equal(mv.view(np.uint32), np.array(list(cam.mv), np.float32).view(np.uint32)), "frame %d: not the orbit's camera" % k
            rout.append({"id": "cfg2_f%d" % k, "mesh": "chessboard.tri", "mode": 6, "w": W, "h": H, "frame": k, "covered": int((tri >= 0).sum()),
                         "overdrawn": int((passes > 1).sum()), "sha256_winners": winners_hash(tri, passes, fat)})
            print(rout[-1], flush=True)
        # ... and at BASELINE config 2's own 1920 x 1080 (oracle/_ref/refraster_1080: the same sources compiled from a temporary copy whose
        # Defines.h:26-27 says 1920 x 1080, SURVEY 8(c)'s recipe)
        if os.path.exists(RC.RASTER_BINARY_1080):
            for k in (0, 37, 100, 150):
                cam, lights, n = O.benchmark_frame(k)
                W, H, mv, tri, passes, fat = RC.raster_winners(osc, 6, list(cam.eye), [0.0, 0.0, 0.0], [list(lights[0].pos)], binary=RC.RASTER_BINARY_1080)
                assert (W, H) == (1920, 1080)
                assert np.array_equal(mv.view(np.uint32), np.array(list(cam.mv), np.float32).view(np.uint32)), "frame %d: not the orbit's camera" % k
                rout.append({"id": "cfg2_1080p_f%d" % k, "mesh": "chessboard.tri", "mode": 6, "w": W, "h": H, "frame": k, "covered": int((tri >= 0).sum()),
                             "overdrawn": int((passes > 1).sum()), "sha256_winners": winners_hash(tri, passes, fat)})
                print(rout[-1], flush=True)
    doc = {"_comment": "Full-size orbit frames beyond f0 from the reference's own Raytracer.cc compiled here (scripts/make_refcore_frame_pins.py: "
                       "oracle/_ref/refcore `raytrace`, strict flags, camera rays of benchmark frame k; f0 made the same way reproduces the survey's "
                       "pins of tests/golden/reference_pins.json).  sha256 over raw R,G,B bytes, row-major, top row first; sha256_f32 over the r,g,b "
                       "float32 values clamped at 255 (what mi355_render hands out as out_rgb_f32).",
           "frames": out,
           "_comment_raster": "oracle/_ref/refraster (the reference's Rasterizers.cc with recording plotters, 800 x 600 = its compile-time size; the cfg2_1080p "
                              "rows: refraster_1080 = the same sources compiled from a temporary copy with Defines.h:26-27 patched to 1920 x 1080) on "
                              "orbit cameras: sha256 over the winning triangle per pixel (int32, -1 = none), the Z-pass count per pixel (int32) and, for "
                              "covered pixels, the bits of the eight floats of the fat point the plotter received (NaN -> 0x7fc00000).",
           "raster_winners": rout}
    json.dump(doc, open(os.path.join(ROOT, "tests", "golden", "refcore_frame_pins.json"), "w"), indent=1)

if __name__ == "__main__":
    main()
