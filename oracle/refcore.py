"""Python side of oracle/_ref/refcore: the SDL-free parts of the REAL reference renderer (oracle/refcore/refcore.cc).

TEST INFRASTRUCTURE ONLY.  The binary is built from /root/reference by oracle/refcore/Makefile (in this container; the GPU
box receives the built file).  It pins the oracle: tests/test_refcore_pins.py runs the same inputs through the reference's
own Raytrace<>, RenderSceneIntoShadowBuffer, Camera / Light bases, LightingEquation<> and MLAA and through the oracle's
restatement, and compares bit for bit.
"""
from __future__ import annotations

import os
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
BINARY = os.path.join(_HERE, "_ref", "refcore")
RASTER_BINARY = os.path.join(_HERE, "_ref", "refraster")
RASTER_BINARY_1080 = os.path.join(_HERE, "_ref", "refraster_1080")      # the same sources with Defines.h:26-27 patched to 1920 x 1080 (refcore/Makefile)
REFERENCE_SRC = "/root/reference/src"


def build() -> str | None:
    """Build the binary where the reference tree is present; returns its path, or None when it cannot exist here."""
    if os.path.isdir(REFERENCE_SRC):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "refcore"), "-s"])
    return BINARY if os.path.exists(BINARY) else None


def available() -> bool:
    return os.path.exists(BINARY)


def _run(cmd: str, blobs, out_dtype, timeout=1800, variant=""):
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in"), os.path.join(d, "out")
        with open(fin, "wb") as f:
            for b in blobs:
                f.write(b if isinstance(b, (bytes, bytearray)) else np.ascontiguousarray(b).tobytes())
        subprocess.run([BINARY + variant, cmd, fin, fout], check=True, timeout=timeout, stdout=subprocess.DEVNULL)
        return np.fromfile(fout, dtype=out_dtype)


def _u32(v):
    return np.uint32(v).tobytes()


def _i32(v):
    return np.int32(v).tobytes()


def scene_blobs(osc):
    """The state after Scene::load, as the oracle's loader exports it (oracle_ctypes.Scene)."""
    vpos, vnrm, vao = osc.vertices()
    t = osc.triangles()
    return [_u32(osc.nv), _u32(osc.nt), vpos, vnrm, vao, t["idx"], t["center"], t["normal"], t["colorf"], t["color32"],
            t["two_sided"], t["plane"]]


def bvh_blobs(osc):
    nodes, tri_idx = osc.bvh()
    return [_u32(len(nodes)), _u32(len(tri_idx)), nodes, tri_idx]


def primary_rays(cam, W, H, SD, xs=None, ys=None, sub=None):
    """Raytracer.cc:563-593 in float32 numpy (one IEEE operation per numpy operation): origins and directions of the
    primary rays of pixels (ys x xs), or of sub-sample `sub` (0..3) of the 4 spp pattern."""
    f = np.float32
    xs = np.arange(W) if xs is None else np.asarray(xs)
    ys = np.arange(H) if ys is None else np.asarray(ys)
    xx, yy = np.meshgrid(xs.astype(f), ys.astype(f))
    if sub is not None:
        xx = xx + f(0.25 - 0.5 * (sub & 1))
        yy = yy + f(0.25 - 0.5 * ((sub & 2) >> 1))
    lx = (f(H // 2) - yy) / f(SD)
    ly = (xx - f(W // 2)) / f(SD)
    lz = np.ones_like(lx)
    n = np.sqrt(lx * lx + ly * ly + lz * lz)
    lx, ly, lz = lx / n, ly / n, lz / n
    mv = np.array(list(cam.mv), f).reshape(3, 3)
    d = [mv[0, k] * lx for k in range(3)]
    d = [d[k] + mv[1, k] * ly for k in range(3)]
    d = [d[k] + mv[2, k] * lz for k in range(3)]
    n = np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
    d = np.stack([d[k] / n for k in range(3)], axis=-1)
    o = np.broadcast_to(np.array(list(cam.eye), f), d.shape)
    return np.concatenate([o, d], axis=-1).reshape(-1, 6).astype(f)


def raytrace(osc, cam, lights, n_lights, rays, max_depth=3, variant=""):
    """Raytrace<true>(origin, dir, NULL, 3 - max_depth) of the reference for every ray -> (n, 3) r,g,b floats.
    (MAX_RAY_DEPTH is a #define of 3 in Raytracer.cc:56; starting the recursion at depth 3 - k leaves k levels.)
    variant "_refr": the build with -DREFRACTIONS (its two refractive indices alternate with the parity of the depth,
    so only max_depth 3 and 1 start on the parity a camera ray has); "_ao": -DAMBIENT_OCCLUSION, rand() from a fresh
    process, rays traced in the order given."""
    lp = np.array([list(lights[i].pos) for i in range(n_lights)], np.float32).reshape(-1)
    blobs = scene_blobs(osc) + bvh_blobs(osc) + [_u32(n_lights), lp, np.array(list(cam.eye), np.float32),
                                                 np.array(list(cam.mv), np.float32), _i32(3 - max_depth), _u32(len(rays)), rays]
    return _run("raytrace", blobs, np.float32, variant=variant).reshape(-1, 3)


OMP_BINARY = os.path.join(_HERE, "_ref", "refcore_omp")
# the same driver compiled with the flags of the reference's own release build (configure.ac:47-50, 193-264; oracle/refcore/Makefile):
# a speed baseline only -- fast-math changes pixels
AUTHOR_BINARY = os.path.join(_HERE, "_ref", "refcore_omp_author")


def timing_available() -> bool:
    return os.path.exists(OMP_BINARY)


def time_frames(osc, cams, lights, n_lights, W, H, SD, threads=1, schedule=1, want_last=False, timeout=1800, binary=None):
    """The reference's Raytrace<true> timed on whole frames (refcore.cc `timeframes`, the binary built with -fopenmp for the
    driver's frame loop): cams = [(eye[3], mv[9]), ...] -> (seconds per frame, last frame's [H, W, 3] r,g,b floats or None).
    schedule 0 = the reference's OpenMP shape (a parallel-for over x per scanline), 1 = one parallel loop over scanlines."""
    lp = np.array([list(lights[i].pos) for i in range(n_lights)], np.float32).reshape(-1)
    cam_rows = np.array([list(c.eye) + list(c.mv) for c in cams], np.float32).reshape(-1)
    blobs = scene_blobs(osc) + bvh_blobs(osc) + [_u32(n_lights), lp, _i32(W), _i32(H), np.float32(SD).tobytes(), _i32(threads),
                                                 _i32(schedule), _u32(len(cams)), cam_rows, _i32(1 if want_last else 0)]
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in"), os.path.join(d, "out")
        with open(fin, "wb") as f:
            for b in blobs:
                f.write(b if isinstance(b, (bytes, bytearray)) else np.ascontiguousarray(b).tobytes())
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), OMP_PROC_BIND="false")
        subprocess.run([binary or OMP_BINARY, "timeframes", fin, fout], check=True, timeout=timeout, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, env=env)
        raw = open(fout, "rb").read()
    secs = np.frombuffer(raw[:8 * len(cams)], np.float64).copy()
    last = np.frombuffer(raw[8 * len(cams):], np.float32).reshape(H, W, 3).copy() if want_last else None
    return secs, last


def shadowmap(osc, light_pos):
    out = _run("shadowmap", scene_blobs(osc) + [np.array(light_pos, np.float32)], np.float32)
    return out[:9].copy(), out[9:].reshape(1024, 1024)


def camera_bases(eyes, lookats, lights):
    rows = np.concatenate([np.asarray(eyes, np.float32), np.asarray(lookats, np.float32), np.asarray(lights, np.float32)], axis=1)
    out = _run("camera", [_u32(len(rows)), rows], np.float32).reshape(len(rows), 30)
    return dict(mv=out[:, :9], in_camera_space=out[:, 9:12], camera_to_light=out[:, 12:21], world_to_light=out[:, 21:30])


def lighting(osc, light_positions, eye, lookat, mode, points):
    """LightingEquation<mode>::ComputePixel for rows (inCameraSpace[3], normal[3], material r,g,b, ao)."""
    lp = np.asarray(light_positions, np.float32).reshape(-1)
    blobs = scene_blobs(osc) + [_u32(len(lp) // 3), lp, np.array(list(eye) + list(lookat), np.float32), _i32(mode),
                                _u32(len(points)), np.asarray(points, np.float32)]
    return _run("lighting", blobs, np.float32).reshape(-1, 3)


def mlaa(pixels):
    h, w = pixels.shape
    return _run("mlaa", [_i32(w), _i32(h), pixels.astype(np.uint32)], np.uint32).reshape(h, w)


def bvh(osc):
    """CreateBVH + CreateCFBVH of the reference (scalar builder) -> (nodes[n, 8] uint32, tri_idx); small meshes only."""
    out = _run("bvh", scene_blobs(osc), np.uint32)
    n, ni = int(out[0]), int(out[1])
    return out[2:2 + 8 * n].reshape(n, 8).copy(), out[2 + 8 * n:2 + 8 * n + ni].astype(np.int32)


def raster_available() -> bool:
    return os.path.exists(RASTER_BINARY)


def raster_winners(osc, mode, eye, lookat, light_positions, timeout=1800, binary=None):
    """The reference's OWN rasterizer (oracle/refcore/refraster.cc: RasterizeScene<T>::DrawTriangles, Filler<>, ScanConverter,
    Screen::RasterizeTriangle, the Z-buffer) with recording plotters, at its compile-time 800 x 600 (binary=RASTER_BINARY_1080: 1920 x 1080): returns
    (W, H, camera matrix[9], winning triangle per pixel or -1 [H, W], Z-passes per pixel [H, W], fat point of the last pass [H, W, 8])."""
    lp = np.asarray(light_positions, np.float32).reshape(-1)
    blobs = scene_blobs(osc) + [_u32(len(lp) // 3), lp, np.array(list(eye) + list(lookat), np.float32)]
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in"), os.path.join(d, "out")
        with open(fin, "wb") as f:
            for b in blobs:
                f.write(b if isinstance(b, (bytes, bytearray)) else np.ascontiguousarray(b).tobytes())
        subprocess.run([binary or RASTER_BINARY, str(mode), fin, fout], check=True, timeout=timeout, stdout=subprocess.DEVNULL)
        raw = np.fromfile(fout, dtype=np.uint32)
    W, H = int(raw[0]), int(raw[1])
    mv = raw[2:11].view(np.float32).copy()
    px = raw[11:].reshape(H, W, 10)
    return W, H, mv, px[..., 0].view(np.int32).copy(), px[..., 1].view(np.int32).copy(), px[..., 2:].view(np.float32).copy()
