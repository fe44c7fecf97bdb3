"""Pin the CPU oracle against outputs of the reference recorded by the survey.

*** SURVEY PROVENANCE ***  The whole reference program cannot be rebuilt here (it needs SDL 1.2 development files the image
lacks, and stand-ins are not allowed), so THESE pins are the frame hashes, counters, camera probes and .bvh statistics the
survey recorded from its own strict single-thread build -- nobody can regenerate them in this container.
scripts/make_reference_pins.sh is the recipe for a machine that has SDL 1.2 (and widens the set to frames f0/f50/f100/f150,
modes 1/4/5/7/10 and full .bvh hashes).  What the image CAN build from the reference's own sources is pinned separately and
bit for bit by tests/test_refcore_pins.py (oracle/_ref/refcore: Raytrace<>, shadow maps, camera / light bases,
LightingEquation<>, the scalar BVH builder, MLAA, and the -DREFRACTIONS / -DAMBIENT_OCCLUSION builds).
If the tests below pass, the restatement reproduces the survey's numbers on every BASELINE.json config's first benchmark frame.
"""
import hashlib
import json
import os

import numpy as np
import pytest

PINS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_pins.json")))


ORBIT_PINS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "refcore_frame_pins.json")))


@pytest.mark.parametrize("pin", ORBIT_PINS["frames"], ids=[p["id"] for p in ORBIT_PINS["frames"]])
def test_oracle_reproduces_the_orbit_frame_pins(oracle, oracle_scene, pin):
    """Full-size frames f37 / f100 / f150 of the two raytrace configurations, config 5's 3840x2160, the 4 spp mode and frames with two
    lights, hashed from the reference's own Raytracer.cc in the build container (scripts/make_refcore_frame_pins.py): the oracle's frame
    and float buffer must hash to the same."""
    s = oracle_scene(pin["mesh"], bvh=True)
    cam, lights, n = oracle.benchmark_frame(pin["frame"], pin.get("second_light", False))
    img, imgf, _ = s.render(pin["mode"], cam, lights, n, oracle.default_opts(pin["w"], pin["h"], max_ray_depth=pin["depth"], threads=os.cpu_count() or 1), want_f32=True)
    rgb = np.stack([(img >> 16) & 255, (img >> 8) & 255, img & 255], axis=-1).astype(np.uint8)
    assert hashlib.sha256(rgb.tobytes()).hexdigest() == pin["sha256"]
    assert hashlib.sha256(np.ascontiguousarray(imgf, dtype=np.float32).tobytes()).hexdigest() == pin["sha256_f32"]


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
def test_oracle_reproduces_the_rasterizer_winner_pins(oracle, oracle_scene, pin):
    """BASELINE configs[1] (chessboard.tri, per-pixel Phong) along the orbit at the reference's own compile-time frame size and at the
    configuration's 1920 x 1080 (refraster_1080: the same sources from a temporary copy with Defines.h:26-27 patched): the winning
    triangle, the Z-pass count and the fat point of every pixel as the reference's Rasterizers.cc hands them to its plotter
    (oracle/_ref/refraster in the build container) -- hashed there, reproduced here by the oracle."""
    s = oracle_scene(pin["mesh"])
    cam, lights, n = oracle.benchmark_frame(pin["frame"])
    _, tri, passes, fat = oracle.raster_winners(s, pin["mode"], cam, lights, n, oracle.default_opts(pin["w"], pin["h"]))
    assert int((tri >= 0).sum()) == pin["covered"] and int((passes > 1).sum()) == pin["overdrawn"]
    assert winners_hash(tri, passes, fat) == pin["sha256_winners"]


def test_benchmark_cameras(oracle):
    c0, lights, n = oracle.benchmark_frame(0)
    assert n == 1
    p = PINS["cameras"]
    # the survey printed 9 significant digits: enough to identify a float32 uniquely
    assert np.array_equal(np.array(c0.eye[:], np.float32), np.array(p["f0"]["eye"], np.float32))
    assert np.array_equal(np.array(c0.mv[:], np.float32), np.array(p["f0"]["mv"], np.float32))
    assert np.array_equal(np.array(lights[0].pos[:], np.float32), np.array(p["light"], np.float32))
    for k in (1, 2):
        ck, _, _ = oracle.benchmark_frame(k)
        assert np.array_equal(np.array(ck.eye[:], np.float32), np.array(p["f%d" % k]["eye"], np.float32))


@pytest.mark.parametrize("mesh", list(PINS["mesh_counts"]))
def test_mesh_counts(oracle_scene, mesh):
    s = oracle_scene(mesh)
    assert [s.nv, s.nt] == PINS["mesh_counts"][mesh]


@pytest.mark.parametrize("mesh", list(PINS["bvh"]))
def test_bvh_pins(oracle, oracle_scene, mesh, tmp_path):
    s = oracle_scene(mesh)
    n = s.bvh_build()          # always rebuild here: this is the SAH-builder pin
    pin = PINS["bvh"][mesh]
    assert n == pin["nodes"]
    f = str(tmp_path / "x.bvh")
    s.bvh_save(f)
    data = open(f, "rb").read()
    assert len(data) == 8 + 32 * n + 4 * s.nt
    h = hashlib.sha256(data).hexdigest()
    assert h.startswith(pin["sha_prefix"]) and h.endswith(pin["sha_suffix"])
    nodes, tri_idx = s.bvh()
    inner = (nodes[:, 6] & 0x80000000) == 0
    # idxLeft == own index + 1 for every inner node
    assert np.array_equal(nodes[inner, 6], np.nonzero(inner)[0].astype(np.uint32) + 1)
    assert sorted(tri_idx.tolist()) == list(range(s.nt))


@pytest.mark.parametrize("pin", PINS["frames"], ids=[p["id"] for p in PINS["frames"]])
def test_frame_pins(oracle, oracle_scene, pin):
    cam, lights, n = oracle.benchmark_frame(0)
    s = oracle_scene(pin["mesh"], bvh=pin["mode"] >= 9)
    o = oracle.default_opts(pin["w"], pin["h"], max_ray_depth=pin["depth"], threads=os.cpu_count() or 1)
    maps = [s.shadowmap(lights[0])] if pin["mode"] in (7, 8) else None
    img, _, st = s.render(pin["mode"], cam, lights, n, o, shadow_maps=maps)
    rgb = oracle.rgb_bytes(img)
    assert int((img != 0).sum()) == pin["nonblack"]
    assert int(np.frombuffer(rgb, np.uint8).sum(dtype=np.int64)) == pin["sum"]
    assert hashlib.sha256(rgb).hexdigest() == pin["sha256"]
    ctr = PINS["counters"].get(pin["id"])
    if ctr:
        got = st.as_dict()
        assert {k: got[k] for k in ctr} == ctr
    if pin["id"] == "cfg2":
        assert st.tris_drawn == PINS["counters"]["raster_cfg2"]["tris_drawn"]
