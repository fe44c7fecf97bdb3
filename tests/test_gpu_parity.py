"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
the pinned CPU oracle and against the reference's own frame hashes.

Bar: bit-exact XRGB words in every mode (integer rasterizer modes AND raytrace modes); the
pre-quantisation float buffer of the raytrace modes within 1e-4 per channel (north_star's
tolerance) -- in practice it is bit-exact too, and the test reports that."""
import hashlib
import json
import os

import numpy as np
import pytest

import renderer_amd as R

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
PINS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_pins.json")))
# full-size orbit frames beyond f0, from the reference's own Raytracer.cc compiled in the build container (scripts/make_refcore_frame_pins.py)
ORBIT_PINS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "refcore_frame_pins.json")))
NCPU = os.cpu_count() or 1
FLOAT_TOL = 1e-4


@pytest.fixture(scope="module")
def gpu_scene():
    cache = {}

    def get(name, bvh=False):
        if name not in cache:
            cache[name] = R.Scene(R.assets.mesh_path(name))
        s = cache[name]
        if bvh and s.bvh_info()[0] == 0:
            s.bvh_create()
        return s
    return get


def both_frames(oracle, oracle_scene, gpu_scene, mesh, mode, W, H, frame=0, second_light=False, want_f32=False,
                **optkw):
    bvh = mode >= 9
    hs, osc = gpu_scene(mesh, bvh), oracle_scene(mesh, bvh)
    cam, lights, n = R.benchmark_frame(frame, second_light)
    ocam, olights, on = oracle.benchmark_frame(frame, second_light)
    ho = R.default_opts(W, H, **optkw)
    oo = oracle.default_opts(W, H, threads=NCPU, **{k: v for k, v in optkw.items() if k not in ("collect_stats", "tune")})
    maps = None
    if mode in (7, 8):
        maps = [osc.shadowmap(olights[i]) for i in range(on)]
        for i in range(n):
            hs.shadowmap_render(i, lights[i])
    g = hs.render(mode, cam, lights, n, ho, want_f32=want_f32)
    o = osc.render(mode, ocam, olights, on, oo, shadow_maps=maps, want_f32=want_f32)
    return g, o


def assert_same(g, o):
    diff = int((g[0] != o[0]).sum())
    assert diff == 0, "%d pixels differ from the oracle" % diff
    if g[1] is not None:
        err = float(np.abs(g[1] - o[1]).max())
        assert err <= FLOAT_TOL, "float frame off by %g" % err


def test_device_float_ops_match_host():
    """Known answers for the float operations every pixel rests on (SURVEY.md 7 "float parity" / "float->int"): IEEE
    division and square root, separately rounded multiply-add (no FMA contraction), denormals kept, and the x86
    cvttss2si casts (out-of-range and NaN -> 0x80000000), on gfx950 against IEEE float32 arithmetic computed on the
    host (numpy: one correctly rounded operation per call; the casts spelled out as the x86 instruction defines them)."""
    import ctypes as C
    rng = np.random.default_rng(99)
    n = 1 << 18
    f = np.float32
    def mix():
        x = rng.uniform(-1, 1, n) * np.exp2(rng.integers(-149, 128, n).astype(np.float64))
        x = x.astype(f)
        edge = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 1.5, 2.5, -2.5, 255.0, 255.5, 256.0, 2147483520.0, 2147483648.0,
                         -2147483648.0, -2147483904.0, 4294967296.0, 1e-45, -1e-45, 1.1754942e-38, 1.17549435e-38, 3.4028235e38,
                         np.inf, -np.inf, np.nan, 0.49999997, -0.49999997, 8388607.5, 8388608.0, 16777216.0, 0.1, 1 / 3], f)
        x[rng.integers(0, n, 4096)] = edge[rng.integers(0, len(edge), 4096)]
        return x
    a, b, c = mix(), mix(), mix()
    a[:n // 4] = np.abs(a[:n // 4]); a[n // 4:n // 2] = rng.uniform(-300, 300, n // 4).astype(f)   # square roots, colour range
    out = np.zeros((9, n), np.uint32)
    rc = R.lib().mi355i_float_kat(C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), C.c_void_p(c.ctypes.data),
                                  C.c_void_p(out.ctypes.data), C.c_uint32(n))
    assert rc == 0, R.lib().mi355_last_error().decode()
    with np.errstate(all="ignore"):
        def cvtt(x):
            ok = (x >= f(-2147483648.0)) & (x < f(2147483648.0))
            return np.where(ok, np.trunc(np.where(ok, x, 0)).astype(np.int64), -2147483648).astype(np.int64).astype(np.uint32)
        prod = a * b
        want = [a / b, np.sqrt(a), prod + c, a + b, prod]
        names = ["div", "sqrt", "mul_then_add", "add", "mul"]
        for k, (w, name) in enumerate(zip(want, names)):
            got = out[k].view(f)
            same = (out[k] == w.view(np.uint32)) | (np.isnan(got) & np.isnan(w))
            assert same.all(), "%s: %d of %d results differ, e.g. %r" % (name, int((~same).sum()), n, (a[~same][0], b[~same][0], c[~same][0]))
        assert np.array_equal(out[5], cvtt(a)), "cvtt_i32"
        assert np.array_equal(out[6], cvtt(np.where(a < 0, a - f(0.5), a + f(0.5)).astype(f))), "myfloor"
        assert np.array_equal(out[7], cvtt(a) & 0xff), "u8cast"
        w = (prod.astype(np.float64) / 255.0).astype(f)
        got = out[8].view(f)
        assert ((out[8] == w.view(np.uint32)) | (np.isnan(got) & np.isnan(w))).all(), "double-promoted division"
    assert (np.abs(a[np.isfinite(a)]) < 1.1754944e-38).sum() > 1000      # denormal operands were in the mix


@pytest.mark.parametrize("pin", PINS["frames"], ids=[p["id"] for p in PINS["frames"]])
def test_reference_frame_hashes(gpu_scene, pin):
    """HIP output hashed directly against the REAL reference's SHA-256 pins (no oracle in the loop)."""
    hs = gpu_scene(pin["mesh"], pin["mode"] >= 9)
    cam, lights, n = R.benchmark_frame(0)
    o = R.default_opts(pin["w"], pin["h"], max_ray_depth=pin["depth"], collect_stats=1)
    if pin["mode"] in (7, 8):
        hs.shadowmap_render(0, lights[0])
    img, _, st = hs.render(pin["mode"], cam, lights, n, o)
    rgb = R.rgb_bytes(img)
    assert int((img != 0).sum()) == pin["nonblack"]
    assert hashlib.sha256(rgb).hexdigest() == pin["sha256"]
    ctr = PINS["counters"].get(pin["id"])
    if ctr:
        got = st.as_dict()
        want = {k: v for k, v in ctr.items() if k != "max_stack"}      # stackless traversal: no stack to measure
        assert {k: got[k] for k in want} == want
    if pin["id"] == "cfg2":
        assert st.tris_drawn == PINS["counters"]["raster_cfg2"]["tris_drawn"]
    if pin["mode"] >= 9:
        # the counting frame above walks in the reference's order; the production kernel (ordered walk, shadow rays
        # handed to idle lanes) must hash to the same reference frame and trace the same rays
        assert hs.walk_info()[0] == 1
        img2, _, st2 = hs.render(pin["mode"], cam, lights, n, R.default_opts(pin["w"], pin["h"], max_ray_depth=pin["depth"]))
        assert hashlib.sha256(R.rgb_bytes(img2)).hexdigest() == pin["sha256"]
        assert (st2.normal_rays, st2.shadow_rays) == (st.normal_rays, st.shadow_rays)


@pytest.mark.parametrize("pin", ORBIT_PINS["frames"], ids=[p["id"] for p in ORBIT_PINS["frames"]])
def test_reference_orbit_frame_hashes(gpu_scene, pin):
    """The raytrace configurations at full size further along the orbit (frames f37, f100, f150), config 5's 3840x2160, the 4 spp mode
    and frames with two lights: the production kernel's packed frame AND its float buffer hash to what the reference's own
    Raytrace<true> gives for those cameras -- no oracle in the loop --, alone, as one of a batch, and from the walk in the reference's
    order."""
    hs = gpu_scene(pin["mesh"], True)
    mode, two = pin["mode"], pin.get("second_light", False)
    cam, lights, n = R.benchmark_frame(pin["frame"], two)
    o = R.default_opts(pin["w"], pin["h"], max_ray_depth=pin["depth"])
    img, imgf, _ = hs.render(mode, cam, lights, n, o, want_f32=True)
    assert int((img != 0).sum()) == pin["nonblack"]
    assert hashlib.sha256(R.rgb_bytes(img)).hexdigest() == pin["sha256"]
    assert hashlib.sha256(np.ascontiguousarray(imgf, dtype=np.float32).tobytes()).hexdigest() == pin["sha256_f32"]
    img2, _, _ = hs.render(mode, cam, lights, n, R.default_opts(pin["w"], pin["h"], max_ray_depth=pin["depth"], tune=R.tune(reforder=1)))
    assert hashlib.sha256(R.rgb_bytes(img2)).hexdigest() == pin["sha256"]
    # the same camera as frame 3 of a batch of 8 (the launch the bench line times)
    import torch
    dev = torch.device("cuda", 0)
    frames = [(pin["frame"] - 3 + j) % 200 for j in range(8)]
    cs = [R.benchmark_frame(f, two) for f in frames]
    outs = [torch.zeros((pin["h"], pin["w"]), dtype=torch.int32, device=dev) for _ in range(8)]
    hs.render_batch_device(mode, [c[0] for c in cs], [c[1] for c in cs], n, o, [t.data_ptr() for t in outs], pin["w"] * 4, None, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    got = outs[3].cpu().numpy().view(np.uint32)
    assert hashlib.sha256(R.rgb_bytes(got)).hexdigest() == pin["sha256"]


def winners_hash(tri, passes, fat):
    """scripts/make_refcore_frame_pins.py: winners_hash"""
    bits = np.ascontiguousarray(fat, np.float32).view(np.uint32).copy()
    bits[np.isnan(fat)] = 0x7fc00000
    bits[tri < 0] = 0
    h = hashlib.sha256()
    for x in (np.ascontiguousarray(tri, np.int32), np.ascontiguousarray(passes, np.int32), bits):
        h.update(x.tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("pin", ORBIT_PINS["raster_winners"], ids=[p["id"] for p in ORBIT_PINS["raster_winners"]])
def test_raster_frames_on_the_pinned_winner_maps(oracle, oracle_scene, gpu_scene, pin):
    """The rasterizer along the orbit at the reference's compile-time 800 x 600 and (round 6) at BASELINE config 2's own 1920 x 1080: the
    oracle's winner maps are the ones the reference's own Rasterizers.cc produced in the build container (hash pin), and the HIP frame is
    the frame the oracle plots from them."""
    osc, hs = oracle_scene(pin["mesh"]), gpu_scene(pin["mesh"])
    ocam, olights, on = oracle.benchmark_frame(pin["frame"])
    oimg, tri, passes, fat = oracle.raster_winners(osc, pin["mode"], ocam, olights, on, oracle.default_opts(pin["w"], pin["h"]))
    assert winners_hash(tri, passes, fat) == pin["sha256_winners"]
    cam, lights, n = R.benchmark_frame(pin["frame"])
    img, _, _ = hs.render(pin["mode"], cam, lights, n, R.default_opts(pin["w"], pin["h"]))
    assert np.array_equal(img, oimg)
    assert int((img != 0).sum()) >= pin["covered"] * 9 // 10        # (a lit picture: nearly every covered pixel has a colour)


@pytest.mark.parametrize("cfg", [
    ("cfg5", "dragon_vis.ply", 9, 3840, 2160, 3), ("statue_depth3_1080p", "statue.ply", 9, 1920, 1080, 3),
    ("chessboard_depth3_1080p", "chessboard.tri", 9, 1920, 1080, 3), ("dragon_aa_1080p", "dragon_vis.ply", 10, 1920, 1080, 3)],
    ids=lambda c: c[0])
def test_full_size_configs_counters_and_oracle(oracle, oracle_scene, gpu_scene, cfg):
    """BASELINE config 5 (4K) and the other full-size rows of SURVEY 8(d): the reference's own counters (counting
    frame, reference order) and the oracle's pixels (production kernel) at full size."""
    key, mesh, mode, W, H, depth = cfg
    hs = gpu_scene(mesh, True)
    cam, lights, n = R.benchmark_frame(0)
    st = hs.render(mode, cam, lights, n, R.default_opts(W, H, max_ray_depth=depth, collect_stats=1))[2].as_dict()
    want = PINS["counters"][key]
    assert {k: st[k] for k in want} == want
    g, o = both_frames(oracle, oracle_scene, gpu_scene, mesh, mode, W, H, 0, want_f32=True, max_ray_depth=depth)
    assert_same(g, o)
    assert (g[2].normal_rays, g[2].shadow_rays) == (want["normal_rays"], want["shadow_rays"])


@pytest.mark.parametrize("mesh", ["dragon_vis.ply", "statue.ply", "chessboard.tri"])
@pytest.mark.parametrize("frame", [0, 37])
def test_raytrace_small_frames(oracle, oracle_scene, gpu_scene, mesh, frame):
    g, o = both_frames(oracle, oracle_scene, gpu_scene, mesh, 9, 480, 270, frame, want_f32=True)
    assert_same(g, o)
    assert g[2].normal_rays == o[2].normal_rays and g[2].shadow_rays == o[2].shadow_rays


@pytest.mark.parametrize("kw", [dict(max_ray_depth=1), dict(max_ray_depth=2), dict(max_ray_depth=4),
                                dict(use_shadows=0), dict(use_reflections=0), dict(use_shadows=0, use_reflections=0)],
                         ids=lambda k: ",".join("%s=%s" % kv for kv in k.items()))
def test_raytrace_option_matrix(oracle, oracle_scene, gpu_scene, kw):
    g, o = both_frames(oracle, oracle_scene, gpu_scene, "dragon_vis.ply", 9, 400, 300, 5, want_f32=True, **kw)
    assert_same(g, o)


def test_raytrace_antialias_and_two_lights(oracle, oracle_scene, gpu_scene):
    g, o = both_frames(oracle, oracle_scene, gpu_scene, "dragon_vis.ply", 10, 320, 240, 3, want_f32=True)
    assert_same(g, o)
    g, o = both_frames(oracle, oracle_scene, gpu_scene, "chessboard.tri", 9, 400, 300, 11, second_light=True,
                       want_f32=True)
    assert_same(g, o)


@pytest.mark.parametrize("mesh", ["dragon_vis.ply", "statue.ply", "chessboard.tri"])
def test_shipped_meshes_take_the_ordered_walk(gpu_scene, mesh):
    """The reference builder's trees pass the checks, so the frames above really ran the ordered kernel."""
    ok, depth, nodes, tame = gpu_scene(mesh, True).walk_info()
    assert ok == 1 and tame == 1 and 1 <= depth <= 48 and nodes > 0


def test_unchecked_tree_falls_back_to_reference_order(oracle, oracle_scene):
    """A tree whose boxes do not contain their triangles (legal input: the reference walks whatever boxes it
    is given) must be walked in the reference's order; one whose boxes are merely larger keeps the ordered walk
    and must give the pixels of the reference-order walk over the same tree."""
    s = R.Scene(R.assets.mesh_path("statue.ply"))
    s.bvh_create()
    nodes, idx = s.bvh_arrays()
    nodes, idx = nodes.copy(), idx.copy()
    cam, lights, n = R.benchmark_frame(4)
    o = R.default_opts(320, 180)
    ref = s.render(9, cam, lights, n, o)[0]
    assert s.walk_info()[0] == 1
    inner = (nodes[:, 6] & 0x80000000) == 0
    f = nodes[:, :6].view(np.float32)
    grown = nodes.copy()
    gf = grown[:, :6].view(np.float32)
    gf[:, :3] -= 0.01
    gf[:, 3:] += 0.01
    s.set_bvh_arrays(grown, idx)
    assert s.walk_info()[0] == 1
    ro = R.default_opts(320, 180, tune=dict(reforder=1))
    assert (s.render(9, cam, lights, n, o)[0] == s.render(9, cam, lights, n, ro)[0]).all()
    # The reference never looks at a LEAF's box (Raytracer.cc:222-230 tests the popped node's own box, inner nodes
    # only), so whatever a tree says there must not change a pixel: larger leaf boxes keep the ordered walk (which
    # may use them only as a conservative filter), meaningless ones send the tree to the reference-order walk.
    leaf = ~inner
    big_leaves = nodes.copy()
    bf = big_leaves[:, :6].view(np.float32)
    bf[leaf, :3] -= 0.05
    bf[leaf, 3:] += 0.05
    s.set_bvh_arrays(big_leaves, idx)
    assert s.walk_info()[0] == 1
    assert (s.render(9, cam, lights, n, o)[0] == ref).all()
    no_leaves = nodes.copy()
    no_leaves[leaf, :6] = 0
    s.set_bvh_arrays(no_leaves, idx)
    assert s.walk_info()[0] == 0
    assert (s.render(9, cam, lights, n, o)[0] == ref).all()
    shrunk = nodes.copy()
    sf = shrunk[:, :6].view(np.float32)
    mid = 0.5 * (f[:, :3] + f[:, 3:])
    sf[inner, :3] = (mid + 0.9 * (f[:, :3] - mid))[inner]
    sf[inner, 3:] = (mid + 0.9 * (f[:, 3:] - mid))[inner]
    s.set_bvh_arrays(shrunk, idx)
    assert s.walk_info()[0] == 0
    a = s.render(9, cam, lights, n, o)[0]
    b = s.render(9, cam, lights, n, ro)[0]
    c = s.render(9, cam, lights, n, R.default_opts(320, 180, collect_stats=1))[0]
    assert (a == b).all() and (a == c).all()


def test_raytrace_stats_variant_matches_and_counts(oracle, oracle_scene, gpu_scene):
    g, o = both_frames(oracle, oracle_scene, gpu_scene, "statue.ply", 9, 640, 360, 2, collect_stats=1)
    assert_same(g, o)
    gs, os_ = g[2].as_dict(), o[2].as_dict()
    for k in ("normal_rays", "shadow_rays", "node_pops", "inner_box_hits", "tri_tests", "plane_pass", "shaded_hits"):
        assert gs[k] == os_[k], k


@pytest.mark.parametrize("size", [(1, 1), (7, 5), (333, 217), (1921, 3)])
def test_raytrace_ragged_sizes(oracle, oracle_scene, gpu_scene, size):
    g, o = both_frames(oracle, oracle_scene, gpu_scene, "dragon_vis.ply", 9, size[0], size[1], 0, want_f32=True)
    assert_same(g, o)


@pytest.mark.parametrize("knobs", [
    dict(xmin=1, rmin=1, chunk=64, lmin=1), dict(xmin=32, rmin=48, chunk=512, lmin=64), dict(xmin=64, rmin=64, chunk=4096),
    dict(exact=1), dict(rowmajor=1), dict(exact=1, lmin=16), dict(scatter=1), dict(bpc=1), dict(bpc=2, exact=1, lmin=4),
    dict(chunk=64, rmin=64, xmin=1, lmin=2), dict(bpc=3), dict(bpc=4), dict(bpc=4, exact=1), dict(reforder=1), dict(reforder=1, exact=1), dict(reforder=1, xmin=1, rmin=1),
    # the work sharing inside a wave (off, eager, late)
    dict(noshare=1), dict(noshare=1, bpc=4), dict(sharemin=1), dict(sharemin=1, bpc=3), dict(sharemin=1, bpc=4), dict(sharemin=32), dict(sharemin=64, bpc=4),
    dict(sharemin=1, exact=1)],
    ids=lambda k: ",".join("%s=%s" % kv for kv in k.items()))
def test_raytrace_tuning_knobs_do_not_change_pixels(oracle, oracle_scene, gpu_scene, knobs):
    """Scheduling knobs and the filtered box test are invisible in the output."""
    g, o = both_frames(oracle, oracle_scene, gpu_scene, "dragon_vis.ply", 9, 640, 360, 9, want_f32=True, tune=knobs)
    assert_same(g, o)
    g, o = both_frames(oracle, oracle_scene, gpu_scene, "statue.ply", 9, 320, 180, 3, want_f32=True, tune=knobs)
    assert_same(g, o)


@pytest.mark.parametrize("mode", [1, 2, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("mesh", ["chessboard.tri", "dragon_vis.ply"])
def test_raster_modes(oracle, oracle_scene, gpu_scene, mesh, mode):
    g, o = both_frames(oracle, oracle_scene, gpu_scene, mesh, mode, 800, 600, 4)
    assert_same(g, o)
    if mode >= 4:
        g, o = both_frames(oracle, oracle_scene, gpu_scene, mesh, mode, 800, 600, 4, collect_stats=1)
        assert_same(g, o)
        assert g[2].tris_drawn == o[2].tris_drawn and g[2].spans == o[2].spans and g[2].ztests == o[2].ztests


@pytest.mark.parametrize("nt", [64, 128, 512])
def test_raster_threads_per_tile_do_not_change_pixels(oracle, oracle_scene, gpu_scene, nt):
    """tune[3] (threads per tile block): 64- and 128-thread blocks lost triangles before round 5 (rs_tile_stage)."""
    for mode in (6, 8):
        g, o = both_frames(oracle, oracle_scene, gpu_scene, "chessboard.tri", mode, 1920, 1080, 0, tune=dict(rsnt=nt))
        assert_same(g, o)


@pytest.mark.parametrize("mode", [1, 2, 4, 5, 6, 7, 8, 9, 10])
def test_model_loaded_from_3ds(oracle, oracle_scene, gpu_scene, mode):
    """legocar.3ds through the host layer's own .3ds reader vs the oracle fed with the REAL lib3ds' dump of the same
    file (tests/test_host_3ds.py): every face has its own three vertices, normals come from smoothing groups, and 120
    faces are two-sided (no back-face culling for them: Raytracer.cc:247, Rasterizers.cc:259)."""
    for frame in (0, 57):
        g, o = both_frames(oracle, oracle_scene, gpu_scene, "legocar.3ds", mode, 640, 480, frame, want_f32=mode >= 9,
                           second_light=mode in (6, 9))
        assert_same(g, o)
    if mode == 9:
        g, o = both_frames(oracle, oracle_scene, gpu_scene, "legocar.3ds", mode, 320, 240, 3, collect_stats=1)
        assert_same(g, o)
        for k in ("normal_rays", "shadow_rays", "node_pops", "tri_tests", "plane_pass"):
            assert getattr(g[2], k) == getattr(o[2], k), k


@pytest.mark.parametrize("mode", [1, 2, 4, 5, 6, 7, 8])
def test_builtin_platform_scene(oracle, mode):
    """The loader's built-in `@platform` square (two triangles, never rescaled) in every raster mode."""
    s, o = R.Scene("@platform"), oracle.Scene("@platform")
    eye, look = np.array([1.1, 0.4, 0.9], np.float32), np.array([0.0, 0.0, 0.0], np.float32)
    cam, ocam = R.camera(eye, look), oracle.camera(eye, look)
    lp = np.array([0.5, -1.0, 2.0], np.float32)
    lights, ol = (R.Light * 2)(R.light(lp, cam)), (oracle.Light * 2)(oracle.light(lp, ocam))
    maps = None
    if mode in (7, 8):
        maps = [o.shadowmap(ol[0])]
        s.shadowmap_render(0, lights[0])
    img = s.render(mode, cam, lights, 1, R.default_opts(320, 240))[0]
    want = o.render(mode, ocam, ol, 1, oracle.default_opts(320, 240), shadow_maps=maps)[0]
    assert np.array_equal(img, want) and img.any()


def test_raster_scratch_regrows_between_frames(oracle, oracle_scene):
    """One context, growing demands: tiny frame, larger frame, shadow map (1024 rows per triangle at most), larger frame
    again.  Every regrowth of the raster scratch must leave the context usable (a double free here once left a sticky
    'invalid argument' behind that failed the next launch)."""
    s = R.Scene(R.assets.mesh_path("chessboard.tri"))
    osc = oracle_scene("chessboard.tri")
    cam, lights, n = R.benchmark_frame(4)
    ocam, ol, on = oracle.benchmark_frame(4)
    for mode, (W, H) in ((6, (64, 48)), (4, (640, 480)), (8, (333, 217)), (6, (1920, 1080)), (7, (800, 600))):
        maps = None
        if mode in (7, 8):
            maps = [osc.shadowmap(ol[0])]
            s.shadowmap_render(0, lights[0])
        img = s.render(mode, cam, lights, n, R.default_opts(W, H))[0]
        want = osc.render(mode, ocam, ol, on, oracle.default_opts(W, H, threads=NCPU), shadow_maps=maps)[0]
        assert np.array_equal(img, want), "mode %d at %dx%d" % (mode, W, H)


def _write_soup(path, verts, faces, cols):
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nelement face %d\nend_header\n" % (len(verts), len(faces)))
        for q in verts:
            f.write("%r %r %r 200\n" % (float(q[0]), float(q[1]), float(q[2])))
        for t, c in zip(faces, cols):
            f.write("3 %d %d %d %d %d %d\n" % (t[0], t[1], t[2], c[0], c[1], c[2]))


def test_few_screen_filling_triangles(oracle, tmp_path):
    """Seven triangles larger than the view, seen from inside the cloud (found by scripts/fuzz_raster.py: the span-chunk
    buffer used to be sized for 4 chunks per row, the dropped triangles' reservations were read as garbage and the GPU
    faulted).  Small scenes now get worst-case buffers."""
    rng = np.random.default_rng(11)
    v = rng.uniform(-1, 1, (7, 1, 3)) + rng.uniform(-2.5, 2.5, (7, 3, 3))
    p = str(tmp_path / "huge.ply")
    _write_soup(p, v.reshape(-1, 3), np.arange(21).reshape(7, 3), rng.integers(30, 255, (7, 3)))
    s, o = R.Scene(p), oracle.Scene(p)
    eye, look = np.array([-0.6, 0.9, 0.4], np.float32), np.array([0.1, 0.0, -0.1], np.float32)
    cam, ocam = R.camera(eye, look), oracle.camera(eye, look)
    lp = np.array([2.0, -1.0, 2.5], np.float32)
    lights, ol = (R.Light * 2)(R.light(lp, cam)), (oracle.Light * 2)(oracle.light(lp, ocam))
    for mode in (4, 5, 6, 7, 8):
        maps = None
        if mode in (7, 8):
            maps = [o.shadowmap(ol[0])]
            s.shadowmap_render(0, lights[0])
        img = s.render(mode, cam, lights, 1, R.default_opts(333, 217))[0]
        want = o.render(mode, ocam, ol, 1, oracle.default_opts(333, 217), shadow_maps=maps)[0]
        assert np.array_equal(img, want), "mode %d" % mode


def test_span_buffers_grow_after_an_overflow(oracle, tmp_path):
    """150 000 triangles (half of them facing the camera) that each cover most of a 200x150 view need more span chunks than
    the default buffer holds:
    mi355_render notices the overflow, doubles the buffers and draws the frame again -- the caller sees a complete
    frame; the asynchronous entry point reports the overflow (-44) and the next frame has the room."""
    rng = np.random.default_rng(5)
    n = 150000
    c = rng.uniform(-0.2, 0.2, (n, 1, 3))
    v = c + rng.uniform(-1.0, 1.0, (n, 3, 3)) * np.array([0.05, 1.0, 1.0])
    p = str(tmp_path / "layers.ply")
    _write_soup(p, v.reshape(-1, 3), np.arange(3 * n).reshape(n, 3), rng.integers(30, 255, (n, 3)))
    s, o = R.Scene(p), oracle.Scene(p)
    eye, look = np.array([2.2, 0.2, 0.1], np.float32), np.array([0.0, 0.0, 0.0], np.float32)
    cam, ocam = R.camera(eye, look), oracle.camera(eye, look)
    lp = np.array([3.0, 1.0, 1.0], np.float32)
    lights, ol = (R.Light * 2)(R.light(lp, cam)), (oracle.Light * 2)(oracle.light(lp, ocam))
    W, H = 200, 150
    dev = torch.device("cuda", 0)
    buf = torch.zeros((H, W), dtype=torch.int32, device=dev)
    s.render_device(4, cam, lights, 1, R.default_opts(W, H), buf.data_ptr(), W * 4, 0, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    with pytest.raises(R.Mi355Error, match="overflowed"):
        s.fetch_stats()
    img = s.render(4, cam, lights, 1, R.default_opts(W, H))[0]          # retries inside until the frame is complete
    want = o.render(4, ocam, ol, 1, oracle.default_opts(W, H))[0]
    assert np.array_equal(img, want)
    s.render_device(4, cam, lights, 1, R.default_opts(W, H), buf.data_ptr(), W * 4, 0, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    s.fetch_stats()                                                      # no overflow any more
    assert np.array_equal(buf.cpu().numpy().astype(np.uint32), want)


@pytest.mark.parametrize("mode", [6, 8])
def test_raster_two_lights_full_hd(oracle, oracle_scene, gpu_scene, mode):
    g, o = both_frames(oracle, oracle_scene, gpu_scene, "chessboard.tri", mode, 1920, 1080, 20, second_light=True)
    assert_same(g, o)


@pytest.mark.parametrize("mesh", ["chessboard.tri", "statue.ply", "dragon_vis.ply"])
def test_shadow_map_bit_exact(oracle, oracle_scene, gpu_scene, mesh):
    hs, osc = gpu_scene(mesh), oracle_scene(mesh)
    cam, lights, n = R.benchmark_frame(0, True)
    _, olights, _ = oracle.benchmark_frame(0, True)
    for i in range(2):
        gm = hs.shadowmap_render(i, lights[i], fetch=True)
        om = osc.shadowmap(olights[i])
        # compare as values: +0 / -0 are interchangeable for every consumer (LightingEq.h:98,113)
        assert np.array_equal(gm, om)
        assert int((gm > -1e30).sum()) > 1000


@pytest.mark.parametrize("mesh,size,pos", [
    ("chessboard.tri", 37, (3.394, 3.394, 4.8)),        # a map smaller than one tile
    ("chessboard.tri", 600, (3.394, 3.394, 4.8)),       # ragged last column of tiles
    ("chessboard.tri", 1025, (-2.0, 3.5, 3.0)),         # one pixel / one row beyond whole tiles and bands
    ("chessboard.tri", 2050, (3.394, 3.394, 4.8)),      # more bands, coarse bands that end early
    ("chessboard.tri", 1024, (0.9, 0.8, 0.7)),          # a light close to the board: triangles many times the map's size, most of them cut
    ("chessboard.tri", 1024, (0.05, 0.02, 0.3)),        # ... above the middle of it: geometry on every side, degenerate projections
    ("dragon_vis.ply", 777, (1.2, -0.9, 0.8)),
    ("statue.ply", 1024, (0.4, 0.3, 0.5)),
    ("chessboard.tri", 4097, (3.394, 3.394, 4.8)),      # large maps: a workgroup per tile, 2049 bands
    ("chessboard.tri", 8192, (3.394, 3.394, 4.8)),      # ... the largest the tile kernels take (SMT_BANDS bands): the lists' numbering at its limit
    ("dragon_vis.ply", 8200, (3.394, 3.394, 4.8)),      # ... and one row more: the row-item kernels take over
], ids=lambda v: str(v).replace(" ", ""))
def test_shadow_map_sizes_and_close_lights(oracle, oracle_scene, gpu_scene, mesh, size, pos):
    """The map's tiles (k_sm_prep / k_sm_tiles: bands of rows, coarse bands for tall triangles, columns from the corners' x) against the
    oracle's serial Light.cc:84-296 where the tiling has edges: sizes that are no multiple of a tile, spans that start far left of
    the map or of a tile (the exact skip-ahead and its bisection), triangles behind and through the light's plane."""
    hs, osc = gpu_scene(mesh), oracle_scene(mesh)
    cam, _, _ = R.benchmark_frame(0)
    ocam, _, _ = oracle.benchmark_frame(0)
    p32 = [np.float32(v) for v in pos]
    gm = hs.shadowmap_render(0, R.light(p32, cam), size=size, fetch=True)
    om = osc.shadowmap(oracle.light(np.array(p32, np.float32), ocam), size=size)
    assert gm.shape == om.shape == (size, size)
    assert np.array_equal(gm, om)           # (as values: +0 / -0 are interchangeable for every consumer, LightingEq.h:98,113)
    assert int((gm > -1e30).sum()) > size


@pytest.mark.parametrize("seed", range(12))
def test_shadow_map_of_random_soups(oracle, tmp_path, seed):
    """Random triangle soups -- tiny to many times the map's size, slivers, exact duplicates, coordinates snapped to grids (equal
    depths, shared edges), flat clouds -- lit from outside, from the edge of and from inside the cloud, at map sizes that are and are
    not multiples of the tiles: the map the LDS-tile kernels draw is the oracle's serial map, bit for bit."""
    rng = np.random.default_rng(1000 + seed)
    n_tri = int(rng.choice([1, 7, 60, 400, 3000, 20000]))
    size = float(rng.choice([0.02, 0.1, 0.5, 2.5]))
    c = rng.uniform(-1, 1, (n_tri, 1, 3))
    v = c + rng.uniform(-size, size, (n_tri, 3, 3))
    if rng.random() < 0.3:
        v[:, :, int(rng.integers(0, 3))] *= 0.02
    snap = [None, None, 0.25, 0.0625][int(rng.integers(0, 4))]
    if snap:
        v = np.round(v / snap) * snap
    if rng.random() < 0.3:
        v = np.concatenate([v, v[: max(1, n_tri // 2)]])
    verts = v.reshape(-1, 3).astype(np.float32)
    faces = np.arange(verts.shape[0]).reshape(-1, 3)
    p = str(tmp_path / ("soup%d.ply" % seed))
    with open(p, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nelement face %d\nend_header\n" % (len(verts), len(faces)))
        for q in verts:
            f.write("%r %r %r 60\n" % (float(q[0]), float(q[1]), float(q[2])))
        for t in faces:
            f.write("3 %d %d %d\n" % (t[0], t[1], t[2]))
    try:
        g = R.Scene(p)
    except R.Mi355Error:
        pytest.skip("degenerate soup: the loader declines it")
    o = oracle.Scene(p)
    cam, _, _ = R.benchmark_frame(0)
    ocam, _, _ = oracle.benchmark_frame(0)
    for trial in range(3):
        lp = (rng.uniform(-1, 1, 3) * float(rng.choice([0.3, 1.2, 4.0]))).astype(np.float32)
        msize = int(rng.choice([64, 257, 1024, 1500]))
        gm = g.shadowmap_render(0, R.light(lp, cam), size=msize, fetch=True)
        om = o.shadowmap(oracle.light(lp, ocam), size=msize)
        assert np.array_equal(gm, om), "seed %d trial %d: %d triangles, light %s, map %d: %d texels differ" % (
            seed, trial, faces.shape[0], lp.tolist(), msize, int((gm != om).sum()))


@pytest.mark.parametrize("band_rows", [8, 15])       # multigpu.BAND_ROWS (tile rows) and a height that straddles tiles
@pytest.mark.parametrize("mode", [9, 6, 2])
def test_band_sharding_reassembles_the_frame(oracle, oracle_scene, gpu_scene, mode, band_rows):
    """Screen bands rendered as 4 'GPUs' and interleaved back == the unsharded frame (multi-GPU layout)."""
    from renderer_amd import multigpu
    mesh, W, H = "dragon_vis.ply", 640, 360
    hs = gpu_scene(mesh, True)
    cam, lights, n = R.benchmark_frame(1)
    full, _, _ = hs.render(mode, cam, lights, n, R.default_opts(W, H))
    parts = []
    for r in range(4):
        o = R.default_opts(W, H, band_rows=band_rows, band_index=r, band_count=4, compact_rows=1)
        img, _, _ = hs.render(mode, cam, lights, n, o)
        assert img.shape[0] == multigpu.rows_of_rank(H, band_rows, 4, r)
        parts.append(img)
    assert np.array_equal(multigpu.assemble_numpy(parts, H, band_rows), full)
    # non-compact: rows of other bands stay black
    o = R.default_opts(W, H, band_rows=band_rows, band_index=1, band_count=4, compact_rows=0)
    img, _, _ = hs.render(mode, cam, lights, n, o)
    ys = np.arange(H)
    mine = (ys // band_rows) % 4 == 1
    assert np.array_equal(img[mine], full[mine]) and not img[~mine].any()


def test_cxx_scene_api_frame(oracle, oracle_scene, gpu_scene):
    """Through mi355::Scene::render* (the reference-shaped C++ API a front-end calls)."""
    mesh = "chessboard.tri"
    hs, osc = gpu_scene(mesh, True), oracle_scene(mesh, True)
    cam, lights, n = R.benchmark_frame(0)
    ocam, olights, on = oracle.benchmark_frame(0)
    eye, lp = list(cam.eye), [list(lights[0].pos)]
    for mode in (2, 6, 8, 9):
        img, _ = hs.render_frame_cxx(mode, 640, 480, eye, [0, 0, 0], lp)
        maps = [osc.shadowmap(olights[0])] if mode == 8 else None
        ref, _, _ = osc.render(mode, ocam, olights, on, oracle.default_opts(640, 480, threads=NCPU), shadow_maps=maps)
        assert np.array_equal(img, ref), "mode %d" % mode


def test_render_cli_runs():
    import subprocess
    out = subprocess.run([R.RENDER_CLI, "-b", "-n", "5", "-m", "9", "-W", "640", "-H", "360",
                          R.assets.mesh_path("dragon_vis.ply")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "Rendering 5 frames in" in out.stdout and "fps" in out.stdout


@pytest.mark.parametrize("mode,mesh", [(9, "dragon_vis.ply"), (8, "chessboard.tri")])
def test_render_cli_pipelined_frames_are_the_synchronous_frames(mode, mesh, tmp_path):
    """render_cli -b (mi355::Scene::renderAsync / renderWait, three canvases in flight: its default) writes the frames of the
    reference's own loop -- one synchronous Scene::render* per pass, -p 1"""
    import subprocess
    for tag, extra in (("sync", ["-p", "1"]), ("pipe", [])):          # (three frames in flight is the default of -b)
        out = subprocess.run([R.RENDER_CLI, "-b", "-n", "7", "-m", str(mode), "-W", "640", "-H", "360", "-o", str(tmp_path / tag)] + extra +
                             [R.assets.mesh_path(mesh)], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
    assert "3 in flight" in out.stdout
    for f in range(1, 8):
        a, b = (tmp_path / ("sync_%04d.ppm" % f)).read_bytes(), (tmp_path / ("pipe_%04d.ppm" % f)).read_bytes()
        assert len(a) == 640 * 360 * 3 + 15 and a == b, "frame %d" % f


def test_errors_are_reported_not_swallowed(gpu_scene):
    hs = R.Scene(R.assets.mesh_path("chessboard.tri"))
    cam, lights, n = R.benchmark_frame(0)
    with pytest.raises(R.Mi355Error, match="set_bvh"):
        hs.render(9, cam, lights, n, R.default_opts(64, 48))
    with pytest.raises(R.Mi355Error, match="shadow map"):
        hs.render(8, cam, lights, n, R.default_opts(64, 48))
    with pytest.raises(R.Mi355Error, match="unknown render mode"):
        hs.render(11, cam, lights, n, R.default_opts(64, 48))
    with pytest.raises(R.Mi355Error, match="wireframe"):
        hs.render(3, cam, lights, n, R.default_opts(4096, 16))         # beyond the sort key's fields
    with pytest.raises(R.Mi355Error):
        hs.render(6, cam, lights, n, R.default_opts(0, 48))


def test_light_rotation_redraws_the_map_in_stream_order(oracle, oracle_scene, gpu_scene):
    """mi355_light_update (the W / Q keys of renderer.cc:410-431 without stopping the frames): six light positions, each followed
    at once by soft-shadowed frames on the same stream -- no synchronisation in between; every frame must be the frame the
    oracle draws with the map of ITS light (the redraw is ordered between the frames that come before and after it)."""
    torch = pytest.importorskip("torch")
    mesh, W, H = "chessboard.tri", 640, 360
    hs, osc = gpu_scene(mesh), oracle_scene(mesh)
    stream = torch.cuda.current_stream()
    n_pos, per = 6, 3
    bufs = [torch.zeros((H, W), dtype=torch.int32, device="cuda:0") for _ in range(n_pos * per)]
    o = R.default_opts(W, H)
    want = []
    for i in range(n_pos):
        a = np.float32(np.pi / 4 + 0.4 * i)
        pos = [np.float32(4.8) * np.cos(a), np.float32(4.8) * np.sin(a), np.float32(4.8)]
        gl = hs.light_update(0, pos, 1024, stream.cuda_stream)
        for j in range(per):
            cam, _, _ = R.benchmark_frame(7 * i + j)
            l = R.light(pos, cam)
            assert np.array_equal(np.array(list(gl.world_to_light), np.float32).view(np.uint32), np.array(list(l.world_to_light), np.float32).view(np.uint32))
            lights = (R.Light * 2)(l)
            hs.render_device(8, cam, lights, 1, o, bufs[i * per + j].data_ptr(), W * 4, 0, stream.cuda_stream)
            ocam, _, _ = oracle.benchmark_frame(7 * i + j)
            want.append((ocam, pos))
    torch.cuda.synchronize()
    hs.fetch_stats()                                   # (reports a row buffer that was too small)
    for k, (ocam, pos) in enumerate(want):
        ol = oracle.light(np.array(pos, np.float32), ocam)
        ref = osc.render(8, ocam, (oracle.Light * 2)(ol), 1, oracle.default_opts(W, H), shadow_maps=[osc.shadowmap(ol)])[0]
        assert np.array_equal(bufs[k].cpu().numpy().view(np.uint32), ref), k
