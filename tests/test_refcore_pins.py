"""Pin the CPU oracle against the REAL reference code, run here.

oracle/_ref/refcore is built from /root/reference (oracle/refcore/Makefile): src/Raytracer.cc, src/Light.cc, src/Camera.cc,
src/MLAA.cc and LightingEq.h, compiled from where they lie with the strict flags of SURVEY.md 8(c) -- every part of the hot
path that does not call into SDL's (absent) library.  Each test runs the same inputs through the reference's own code
and through the oracle's restatement and demands bit-identical floats:

  a2-a5  RayIntersectsBox, BVH_IntersectTriangles<>, Raytrace<>       full-size frames of BASELINE configs 3 and 4, more cameras
  a6     4 spp accumulation                                            the four sub-sample rays of mode 10
  b4,b8  ScanConverter + Light::RenderSceneIntoShadowBuffer            three meshes x two lights
  b7     LightingEquation<NoShadows|ShadowMapping|SoftShadowMapping>   200 k points
  b9     Camera::UpdateMV, the three Light:: bases                     the whole benchmark orbit
  b2-b5  RasterizeScene<T>::DrawTriangles, Filler<T> x 5, ScanConverter in the screen's edge order, Screen::RasterizeTriangle
         and the Z-buffer (oracle/_ref/refraster: Rasterizers.cc itself, with plotters that record)      3 meshes x 4 cameras x 5 types

What the reference cannot run without SDL (Scene::load's Triangle constructor, the BVH builder's progress report, the
body of the rasterizer's Plot<> -- a cast per channel, or un-projection + normalisation in front of the pinned ComputePixel --
and SDL_MapRGB's byte packing) stays pinned by the survey's hashes only (tests/test_oracle_pins.py).
On a box without the reference tree the prebuilt binary is used; without either the tests skip.
"""
import os

import numpy as np
import pytest

from oracle import refcore as RC

pytestmark = pytest.mark.skipif(RC.build() is None, reason="oracle/_ref/refcore not built and no reference tree here")


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _same(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return bool(((_bits(a) == _bits(b)) | (np.isnan(a) & np.isnan(b))).all())


@pytest.mark.parametrize("mesh,w,h,depth,frame,two", [
    ("dragon_vis.ply", 1920, 1080, 3, 0, False),      # BASELINE configs[3], the headline workload
    ("statue.ply", 1920, 1080, 1, 0, False),          # BASELINE configs[2]
    ("statue.ply", 640, 360, 3, 50, False),
    ("chessboard.tri", 640, 360, 3, 100, True),
    ("dragon_vis.ply", 640, 360, 2, 150, True),
    ("dragon_vis.ply", 33, 17, 3, 7, False),
], ids=["cfg4_dragon_1080p", "cfg3_statue_1080p_depth1", "statue_f50", "chessboard_f100_2lights", "dragon_f150_depth2_2lights", "ragged"])
def test_raytrace_floats_equal_the_reference(oracle, oracle_scene, mesh, w, h, depth, frame, two):
    s = oracle_scene(mesh, bvh=True)
    cam, lights, n = oracle.benchmark_frame(frame, two)
    o = oracle.default_opts(w, h, max_ray_depth=depth, threads=os.cpu_count() or 1)
    img, imgf, _ = s.render(9, cam, lights, n, o, want_f32=True)
    ref = RC.raytrace(s, cam, lights, n, RC.primary_rays(cam, w, h, 2 * h), depth).reshape(h, w, 3)
    assert (imgf.sum(-1) > 0).sum() > w * h // 50
    assert _same(ref, imgf)
    # ... and the packed frame is the truncation of exactly those floats (Raytracer.cc:600-604)
    c = np.minimum(ref, np.float32(255.0)).astype(np.uint32)
    assert np.array_equal((c[..., 0] << 16) | (c[..., 1] << 8) | c[..., 2], img)


@pytest.mark.parametrize("schedule,threads", [(0, 1), (0, 4), (1, 4)], ids=["per_scanline_1t", "per_scanline_4t", "rows_4t"])
def test_cpu_baseline_driver_traces_the_pinned_frames(oracle, oracle_scene, schedule, threads):
    """bench.py's `cpu_baseline` (kind "reference") times oracle/_ref/refcore_omp `timeframes`: the reference's Raytrace<true>
    under the driver's own frame loop (OpenMP over pixels).  What it traces must be the frames everything else is pinned to:
    its floats equal the `raytrace` command's on the python-made primary rays, for both loop shapes and any thread count."""
    if not RC.timing_available():
        pytest.skip("oracle/_ref/refcore_omp not built")
    s = oracle_scene("dragon_vis.ply", bvh=True)
    w, h = 160, 90
    cams = [oracle.benchmark_frame(f) for f in (0, 37)]
    lights, n = cams[0][1], cams[0][2]
    secs, last = RC.time_frames(s, [c[0] for c in cams], lights, n, w, h, 2 * h, threads=threads, schedule=schedule, want_last=True)
    assert secs.shape == (2,) and (secs > 0).all()
    ref = RC.raytrace(s, cams[1][0], lights, n, RC.primary_rays(cams[1][0], w, h, 2 * h), 3).reshape(h, w, 3)
    assert (ref.sum(-1) > 0).sum() > w * h // 50
    assert _same(ref, last)


def test_antialiased_frame_is_the_sum_of_the_reference_subsamples(oracle, oracle_scene):
    w, h = 320, 180
    s = oracle_scene("dragon_vis.ply", bvh=True)
    cam, lights, n = oracle.benchmark_frame(2)
    o = oracle.default_opts(w, h, threads=os.cpu_count() or 1)
    _, imgf, _ = s.render(10, cam, lights, n, o, want_f32=True)
    acc = np.zeros((h, w, 3), np.float32)
    for sub in (3, 2, 1, 0):                                   # while(pixelsTraced--), Raytracer.cc:570-597
        acc = acc + RC.raytrace(s, cam, lights, n, RC.primary_rays(cam, w, h, 2 * h, sub=sub), 3).reshape(h, w, 3)
    acc = np.minimum(acc / np.float32(4.0), np.float32(255.0))
    assert _same(acc, np.minimum(imgf, np.float32(255.0)))


@pytest.mark.parametrize("mesh", ["chessboard.tri", "dragon_vis.ply", "statue.ply"])
def test_shadow_map_equals_the_reference(oracle, oracle_scene, mesh):
    s = oracle_scene(mesh)
    _, lights, n = oracle.benchmark_frame(0, True)
    for i in range(n):
        w2l, ref = RC.shadowmap(s, list(lights[i].pos))
        assert _same(ref, s.shadowmap(lights[i]))
        assert _same(w2l, list(lights[i].world_to_light))
        assert (ref > -1e30).sum() > 10000                    # the model was drawn into it


def test_camera_and_light_bases_equal_the_reference(oracle):
    eyes, looks, lps, mine = [], [], [], []
    for k in range(200):
        cam, lights, n = oracle.benchmark_frame(k, True)
        for i in range(n):
            eyes.append(list(cam.eye)); looks.append([0.0, 0.0, 0.0]); lps.append(list(lights[i].pos))
            mine.append(list(cam.mv) + list(lights[i].in_camera_space) + list(lights[i].camera_to_light) + list(lights[i].world_to_light))
    rng = np.random.default_rng(11)
    for _ in range(200):                                       # arbitrary cameras, through the same helpers the tests use
        eye, look, lp = rng.uniform(-6, 6, 3), rng.uniform(-1, 1, 3), rng.uniform(-6, 6, 3)
        cam = oracle.camera(eye, look)
        l = oracle.light(lp, cam)
        eyes.append(list(cam.eye)); looks.append(list(np.float32(look))); lps.append(list(l.pos))
        mine.append(list(cam.mv) + list(l.in_camera_space) + list(l.camera_to_light) + list(l.world_to_light))
    b = RC.camera_bases(eyes, looks, lps)
    ref = np.concatenate([b["mv"], b["in_camera_space"], b["camera_to_light"], b["world_to_light"]], axis=1)
    assert _same(ref, np.array(mine, np.float32))


@pytest.mark.parametrize("mode", [0, 1, 2], ids=["no_shadows", "shadow_maps", "soft_shadow_maps"])
def test_lighting_equation_equals_the_reference(oracle, oracle_scene, mode):
    s = oracle_scene("chessboard.tri")
    cam, lights, n = oracle.benchmark_frame(3, True)
    maps = [s.shadowmap(lights[i]) for i in range(n)] if mode else None
    rng = np.random.default_rng(5 + mode)
    N = 200000
    pts = np.empty((N, 10), np.float32)
    z = rng.uniform(3.0, 6.5, N)
    pts[:, 2] = z
    pts[:, 0] = rng.uniform(-0.3, 0.3, N) * z
    pts[:, 1] = rng.uniform(-0.5, 0.5, N) * z
    nr = rng.normal(size=(N, 3))
    pts[:, 3:6] = nr / np.linalg.norm(nr, axis=1, keepdims=True)
    pts[:, 6:9] = rng.integers(0, 256, (N, 3))
    pts[:, 9] = rng.integers(0, 256, N)
    pts[:100, 0:3] = 0                                         # degenerate points: the NaN paths
    mine = oracle.lighting(lights, n, oracle.default_opts(1920, 1080), mode, pts, maps)
    ref = RC.lighting(s, [list(lights[i].pos) for i in range(n)], list(cam.eye), [0.0, 0.0, 0.0], mode, pts)
    assert _same(ref, mine)
    assert 0.2 < float((mine.sum(1) > 100).mean()) < 0.9


def test_bvh_builder_equals_the_reference_on_small_soups(oracle, tmp_path):
    """a8: CreateBVH / Recurse (BVH.cc:96-371, scalar variant) + Scene::CreateCFBVH (Raytracer.cc:651-718) of the
    reference against the oracle's restatement: same nodes, same triangle list.  The reference's builder reports its
    progress through SDL every 65536 candidate planes, so only meshes that stay below that run here (<= ~100
    triangles: ties, duplicates, flat and thin axes, leaves of 1-3 triangles, depth up to ~8)."""
    done = 0
    for it in range(60):
        rng = np.random.default_rng(424242 + it)
        n_tri = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 16, 30, 60, 90]))
        snap = [None, None, 0.5, 0.125, 0.03125][int(rng.integers(0, 5))]
        stretch = np.array([1.0, 1.0, 1.0]) if rng.random() < 0.6 else rng.choice([0.0, 0.05, 1.0, 8.0], 3)
        if not stretch.any():
            stretch[0] = 1.0
        v = (rng.uniform(-1, 1, (n_tri, 1, 3)) + rng.uniform(-0.15, 0.15, (n_tri, 3, 3))) * stretch
        if snap:
            v = np.round(v / snap) * snap
        if rng.random() < 0.3:
            v = np.concatenate([v, v[: max(1, n_tri // 3)]])
        verts = v.reshape(-1, 3)
        p = str(tmp_path / ("b%d.ply" % it))
        with open(p, "w") as f:
            f.write("ply\nformat ascii 1.0\nelement vertex %d\nelement face %d\nend_header\n" % (len(verts), len(verts) // 3))
            for q in verts:
                f.write("%r %r %r 150\n" % (float(q[0]), float(q[1]), float(q[2])))
            for t in range(len(verts) // 3):
                f.write("3 %d %d %d\n" % (3 * t, 3 * t + 1, 3 * t + 2))
        o = oracle.Scene(p)
        if not np.isfinite(o.vertices()[0]).all():
            continue                                   # collapsed to a point by the snapping: the rescale makes NaNs of it
        o.bvh_build()
        nodes, idx = o.bvh()
        rn, ri = RC.bvh(o)
        assert np.array_equal(rn, nodes) and np.array_equal(ri, idx), "soup %d (%d triangles)" % (it, len(verts) // 3)
        done += 1
    assert done >= 40


def test_mlaa_equals_the_reference(oracle, oracle_scene):
    """f4: the oracle's scalar restatement of MLAA.cc (SSE in the reference) against MLAA.cc itself, compiled by
    oracle/refcore: rendered frames and noise (which exercises every quirk of its aligned four-pixel scans)."""
    s = oracle_scene("chessboard.tri")
    cam, lights, n = oracle.benchmark_frame(5)
    for (W, H) in ((800, 600), (1920, 1080), (64, 48), (8, 8), (12, 8)):
        img, _, _ = s.render(6, cam, lights, n, oracle.default_opts(W, H))
        ref = RC.mlaa(img)
        assert np.array_equal(oracle.mlaa(img), ref), "%dx%d" % (W, H)
        assert W < 64 or (ref != img).sum() > 100
    rng = np.random.default_rng(3)
    for it in range(60):
        W, H = int(rng.integers(2, 40)) * 4, int(rng.integers(1, 20)) * 8
        kind = it % 4
        if kind == 0:
            img = rng.integers(0, 1 << 24, (H, W), dtype=np.uint32)
        elif kind == 1:
            img = (rng.integers(0, 3, (H, W)) * 0x404040).astype(np.uint32)
        elif kind == 2:
            img = np.zeros((H, W), np.uint32)
            for _ in range(6):
                x0, y0 = int(rng.integers(0, W)), int(rng.integers(0, H))
                img[y0:y0 + int(rng.integers(1, H)), x0:x0 + int(rng.integers(1, W))] = int(rng.integers(0, 1 << 24))
        else:
            yy, xx = np.mgrid[0:H, 0:W]
            img = (((xx * int(rng.integers(1, 5)) + yy * int(rng.integers(1, 5))) // int(rng.integers(3, 17))) % 2 * 0xffffff).astype(np.uint32)
        assert np.array_equal(oracle.mlaa(img), RC.mlaa(img)), "case %d (%dx%d)" % (it, W, H)
    with pytest.raises(RuntimeError):
        oracle.mlaa(np.zeros((10, 16), np.uint32))


@pytest.mark.parametrize("mesh,w,h,depth,frame,two", [
    ("chessboard.tri", 480, 270, 3, 0, True),
    ("dragon_vis.ply", 480, 270, 3, 40, False),
    ("statue.ply", 320, 180, 1, 90, False),
], ids=["chessboard_2lights", "dragon", "statue_depth1"])
def test_refractions_equal_the_reference_built_with_them(oracle, oracle_scene, mesh, w, h, depth, frame, two):
    """-DREFRACTIONS (Raytracer.cc:72, 526-551): every hit also spawns an unculled refracted ray; the binary tree of
    rays, the alternating indices and the two clamping additions, float for float."""
    s = oracle_scene(mesh, bvh=True)
    cam, lights, n = oracle.benchmark_frame(frame, two)
    o = oracle.default_opts(w, h, max_ray_depth=depth, threads=os.cpu_count() or 1, use_refractions=1)
    _, imgf, _ = s.render(9, cam, lights, n, o, want_f32=True)
    ref = RC.raytrace(s, cam, lights, n, RC.primary_rays(cam, w, h, 2 * h), depth, variant="_refr").reshape(h, w, 3)
    plain = RC.raytrace(s, cam, lights, n, RC.primary_rays(cam, w, h, 2 * h), depth).reshape(h, w, 3)
    assert _same(ref, imgf)
    if depth > 1:
        assert (ref != plain).any(axis=-1).sum() > w * h // 100       # (the option does change the picture)


@pytest.mark.parametrize("mesh,w,h,frame", [("chessboard.tri", 160, 90, 0), ("dragon_vis.ply", 128, 72, 40)])
def test_raycast_ambient_occlusion_equals_the_reference_on_its_rand_sequence(oracle, oracle_scene, mesh, w, h, frame):
    """-DAMBIENT_OCCLUSION (Raytracer.cc:386-417) draws from rand(): one thread tracing the same rays in the same order
    as the reference binary (a fresh process: seed 1) reproduces its floats."""
    s = oracle_scene(mesh, bvh=True)
    cam, lights, n = oracle.benchmark_frame(frame)
    o = oracle.default_opts(w, h, threads=1, ambient_occlusion=1)
    _, imgf, _ = s.render(9, cam, lights, n, o, want_f32=True)
    ref = RC.raytrace(s, cam, lights, n, RC.primary_rays(cam, w, h, 2 * h), 3, variant="_ao").reshape(h, w, 3)
    assert (imgf.sum(-1) > 0).sum() > w * h // 50
    assert _same(ref, imgf)
    # the device's counter-based generator gives the same picture up to sampling noise
    o2 = oracle.default_opts(w, h, threads=os.cpu_count() or 1, ambient_occlusion=2)
    _, img2, _ = s.render(9, cam, lights, n, o2, want_f32=True)
    lit = imgf.sum(-1) > 0
    assert np.array_equal(lit, img2.sum(-1) > 0)
    a, b = np.minimum(imgf, 255.0)[lit].mean(), np.minimum(img2, 255.0)[lit].mean()
    assert abs(a - b) < 0.01 * max(a, b), (a, b)


# ---- the rasterizer itself: rows b2 - b5 against the reference's own code (oracle/refcore/refraster.cc) -------------------
_RASTER_CAMERAS = {
    # (eye, lookat): the benchmark orbit's first camera; one from above and behind; one INSIDE the model's box looking along it
    # (triangles cut by the near distance, spans clipped at both screen edges, edges that start above the screen); one far away
    "orbit0": ([4.799934387207031, -0.02513263002038002, 0.0], [0.0, 0.0, 0.0]),
    "above": ([-2.1, 1.7, 3.3], [0.2, -0.1, 0.0]),
    "inside": ([0.35, 0.1, 0.22], [-1.0, 0.4, -0.1]),
    "far": ([31.0, -17.0, 9.0], [0.0, 0.0, 0.0]),
}


@pytest.mark.parametrize("mesh", ["chessboard.tri", "dragon_vis.ply", "legocar.3ds"])
@pytest.mark.parametrize("mode", [4, 5, 6, 7, 8])
def test_rasterizer_equals_the_reference_up_to_the_plotter(oracle, oracle_scene, mesh, mode):
    """RasterizeScene<T>::DrawTriangles + Filler<T> + ScanConverter + Screen::RasterizeTriangle of the REFERENCE, compiled here,
    with plotters that record what they are given: for every pixel of an 800 x 600 frame the oracle must name the same winning
    triangle, count the same number of Z-passes and hand its plotter the same interpolated fat point, bit for bit -- for all five
    fat-point types (two lights: Gouraud lights its vertices with both)."""
    if not RC.raster_available():
        pytest.skip("oracle/_ref/refraster not built")
    s = oracle_scene(mesh)
    lp = [[3.4, 3.4, 4.8], [-2.5, 1.0, 3.0]]
    seen_lit = seen_multi = 0
    for name, (eye, look) in _RASTER_CAMERAS.items():
        W, H, mv, tri, passes, fat = RC.raster_winners(s, mode, eye, look, lp)
        cam = oracle.camera(np.array(eye, np.float32), np.array(look, np.float32))
        assert np.array_equal(_bits(mv), _bits(list(cam.mv))), name
        lights = (oracle.Light * 2)(*[oracle.light(np.array(p, np.float32), cam) for p in lp])
        maps = [s.shadowmap(lights[i]) for i in range(2)] if mode in (7, 8) else None
        _, otri, opasses, ofat = oracle.raster_winners(s, mode, cam, lights, 2, oracle.default_opts(W, H), shadow_maps=maps)
        assert np.array_equal(tri, otri), "%s: %d pixels won by another triangle" % (name, int((tri != otri).sum()))
        assert np.array_equal(passes, opasses), "%s: Z-pass counts differ at %d pixels" % (name, int((passes != opasses).sum()))
        bad = (_bits(fat) != _bits(ofat)) & ~(np.isnan(fat) & np.isnan(ofat))
        assert not bad.any(), "%s: %d fat-point words differ" % (name, int(bad.sum()))
        seen_lit += int((tri >= 0).sum()); seen_multi += int((passes > 1).sum())
    assert seen_lit > 50000 and seen_multi > 2000        # (pictures, with overdraw)
