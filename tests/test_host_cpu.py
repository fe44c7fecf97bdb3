"""CPU tests of the C++ host layer (no GPU): loaders, fix_normals, rescale, triangle precompute,
BVH builder + .bvh cache, benchmark cameras -- all bit-exact against the pinned oracle -- and the
C-ABI library's exported symbols."""
import ctypes as C
import hashlib
import json
import os
import re
import shutil

import numpy as np
import pytest

import renderer_amd as R

MESHES = ["chessboard.tri", "statue.ply", "dragon_vis.ply"]
PINS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_pins.json")))


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.fixture(scope="module")
def host_scene():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = R.Scene(R.assets.mesh_path(name))
        return cache[name]
    return get


def test_abi_exports_every_declared_symbol():
    """include/mi355_render.h is the contract: every function it declares must be exported."""
    hdr = open(R.HEADER).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(mi355_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 12
    L = R.lib()
    for n in names:
        assert hasattr(L, n), "libmi355render.so does not export %s" % n
    assert L.mi355_abi_version() == 2


def test_struct_sizes_match_header():
    # sizeof() as the C compiler sees them (computed from the header's field lists)
    assert C.sizeof(R.Camera) == 48 and C.sizeof(R.Light) == 96
    assert C.sizeof(R.Opts) == 34 * 4
    assert C.sizeof(R.Stats) == 11 * 8 + 8
    assert C.sizeof(R.SceneDesc) == 8 + 11 * 8


def test_struct_field_offsets_match_header(tmp_path):
    """Every field of the ctypes mirrors lies where the C compiler puts the header's field of the same name (a field added to
    mi355_opts -- keep_canvas took a reserved word in round 6 -- must not shift its neighbours in one of the two)."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("no C compiler")
    structs = (("mi355_opts", R.Opts), ("mi355_camera", R.Camera), ("mi355_light", R.Light), ("mi355_stats", R.Stats))
    src = ['#include <stddef.h>', '#include <stdio.h>', '#include "mi355_render.h"', 'int main(void) {']
    for cname, T in structs:
        src.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for f in T._fields_:
            src.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, f[0], cname, f[0]))
    src += ['return 0;', '}']
    c = tmp_path / "offsets.c"
    c.write_text("\n".join(src))
    exe = str(tmp_path / "offsets")
    subprocess.run(["gcc", "-I", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include"), str(c), "-o", exe], check=True)
    got = dict(l.split() for l in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.splitlines())
    for cname, T in structs:
        assert int(got[cname]) == C.sizeof(T), cname
        for f in T._fields_:
            assert int(got["%s.%s" % (cname, f[0])]) == getattr(T, f[0]).offset, "%s.%s" % (cname, f[0])


def test_no_gpu_means_loud_failure():
    """Without a GPU the library must refuse to render: there is no CPU fallback."""
    if R.device_count() > 0:
        pytest.skip("a GPU is present")
    s = R.Scene(R.assets.mesh_path("chessboard.tri"))
    cam, lights, n = R.benchmark_frame(0)
    with pytest.raises(R.Mi355Error):
        s.render(2, cam, lights, n, R.default_opts(64, 48))


@pytest.mark.parametrize("mesh", MESHES)
def test_loader_and_precompute_bit_exact(oracle_scene, host_scene, mesh):
    o = oracle_scene(mesh)
    h = host_scene(mesh)
    assert [h.nv, h.nt] == PINS["mesh_counts"][mesh]
    vpos, vnrm, vao = o.vertices()
    t = o.triangles()
    a = h.arrays()
    assert np.array_equal(bits(a["vertex_pos"]), bits(vpos))
    assert np.array_equal(bits(a["vertex_normal"]), bits(vnrm))
    assert np.array_equal(a["vertex_ao"], vao)
    assert np.array_equal(a["tri_index"], t["idx"])
    assert np.array_equal(bits(a["tri_center"]), bits(t["center"]))
    assert np.array_equal(bits(a["tri_normal"]), bits(t["normal"]))
    assert np.array_equal(bits(a["tri_colorf"]), bits(t["colorf"]))
    assert np.array_equal(a["tri_color32"], t["color32"])
    assert np.array_equal(a["tri_two_sided"], t["two_sided"])
    assert np.array_equal(bits(a["tri_d"]), bits(t["plane"][:, 0:4]))
    assert np.array_equal(bits(a["tri_e"]), bits(t["plane"][:, 4:13]))


@pytest.mark.parametrize("mesh", MESHES)
def test_bvh_builder_reproduces_reference_tree(host_scene, mesh, tmp_path):
    """The sorted-sweep builder must emit the reference's exact tree: same .bvh bytes (hash pinned)."""
    h = host_scene(mesh)
    n = h.bvh_create("host")
    pin = PINS["bvh"][mesh]
    assert n == pin["nodes"]
    nodes, idx = h.bvh_arrays()
    blob = np.array([n, h.nt], np.uint32).tobytes() + nodes.tobytes() + idx.tobytes()
    assert len(blob) == 8 + 32 * n + 4 * h.nt
    sha = hashlib.sha256(blob).hexdigest()
    assert sha.startswith(pin["sha_prefix"]) and sha.endswith(pin["sha_suffix"])


def test_bvh_cache_roundtrip_and_fallback(tmp_path):
    src = R.assets.mesh_path("dragon_vis.ply")
    model = str(tmp_path / "dragon_vis.ply")
    shutil.copy(src, model)
    s = R.Scene(model)
    n = s.bvh_update()                      # builds and writes <model>.bvh (Raytracer.cc:747-753)
    cache = model + ".bvh"
    assert os.path.getsize(cache) == 8 + 32 * n + 4 * s.nt
    ref = open(cache, "rb").read()
    s2 = R.Scene(model)
    assert s2.bvh_update() == n             # reads the cache
    nodes, idx = s2.bvh_arrays()
    assert ref[8:8 + 32 * n] == nodes.tobytes() and ref[8 + 32 * n:] == idx.tobytes()
    open(cache, "wb").write(ref[:1000])     # short file -> silent rebuild (Raytracer.cc:756-785)
    s3 = R.Scene(model)
    assert s3.bvh_update() == n
    nodes3, idx3 = s3.bvh_arrays()
    assert nodes3.tobytes() == nodes.tobytes() and idx3.tobytes() == idx.tobytes()


def test_benchmark_orbit_matches_oracle_and_reference(oracle):
    for k in (0, 1, 2, 57, 199):
        ch, lh, nh = R.benchmark_frame(k)
        co, lo, no = oracle.benchmark_frame(k)
        assert nh == no == 1
        assert bytes(ch) == bytes(co)
        assert bytes(lh[0]) == bytes(lo[0])
    ch, lh, nh = R.benchmark_frame(3, second_light=True)
    co, lo, no = oracle.benchmark_frame(3, True)
    assert nh == no == 2 and bytes(lh[1]) == bytes(lo[1])
    c0, l0, _ = R.benchmark_frame(0)
    p = PINS["cameras"]
    assert np.array_equal(np.array(c0.eye[:], np.float32), np.array(p["f0"]["eye"], np.float32))
    assert np.array_equal(np.array(c0.mv[:], np.float32), np.array(p["f0"]["mv"], np.float32))
    assert np.array_equal(np.array(l0[0].pos[:], np.float32), np.array(p["light"], np.float32))


def test_malformed_inputs_raise(tmp_path):
    bad = tmp_path / "bad.tri"
    bad.write_bytes(b"\xde\xc0\xad\xde" + b"\x05\x00\x00\x00" + b"\x00" * 10)   # truncated
    with pytest.raises(R.Mi355Error):
        R.Scene(str(bad))
    with pytest.raises(R.Mi355Error):
        R.Scene(str(tmp_path / "missing.ply"))
    with pytest.raises(R.Mi355Error):
        R.Scene(str(tmp_path / "noext"))
    ply = tmp_path / "idx.ply"
    ply.write_text("ply\nelement vertex 3\nelement face 1\nend_header\n0 0 0 1\n1 0 0 1\n0 1 0 1\n3 0 1 7\n")
    with pytest.raises(R.Mi355Error):
        R.Scene(str(ply))


def test_tiny_ply_with_and_without_colours(tmp_path, oracle):
    ply = tmp_path / "t.ply"
    ply.write_text("ply\nformat ascii 1.0\nelement vertex 4\nelement face 2\nend_header\n"
                   "0 0 0 10\n1 0 0 20\n0 1 0 30\n0 0 1 300\n3 0 1 2 200 100 50\n3 0 2 3\n")
    h = R.Scene(str(ply))
    o = oracle.Scene(str(ply))
    a, t = h.arrays(), o.triangles()
    assert a["vertex_ao"].tolist() == [10, 20, 30, 300 & 0xff] == o.vertices()[2].tolist()
    assert a["tri_color32"].tolist() == [0xC86432, 0xFFFFFF] == t["color32"].tolist()
    assert np.array_equal(bits(a["vertex_normal"]), bits(o.vertices()[1]))
    assert np.array_equal(bits(a["tri_e"]), bits(t["plane"][:, 4:13]))
    assert h.bvh_create() == 1      # < 4 triangles: a single leaf


@pytest.mark.timeout(120)
def test_mesh_without_extent_is_refused_not_hung(tmp_path):
    """A mesh collapsed to a point becomes NaN in the loader's rescale (division by its extent, Loader.cc:418-454); the
    reference's plane loop (BVH.cc:154) then never ends.  The builders here refuse such a scene instead."""
    ply = tmp_path / "point.ply"
    ply.write_text("ply\nformat ascii 1.0\nelement vertex 15\nelement face 5\nend_header\n" + "1 2 3 10\n" * 15 +
                   "".join("3 %d %d %d\n" % (3 * i, 3 * i + 1, 3 * i + 2) for i in range(5)))
    s = R.Scene(str(ply))
    assert not np.isfinite(s.arrays()["vertex_pos"]).all()
    with pytest.raises(R.Mi355Error, match="non-finite"):
        s.bvh_create("host")


def test_builtin_platform_scene(oracle):
    """`@p...` is the loader's built-in unit square (Loader.cc:87-97); it skips the loader's common tail."""
    h, o = R.Scene("@platform"), oracle.Scene("@platform")
    assert (h.nv, h.nt) == (o.nv, o.nt) == (4, 2)
    a = h.arrays()
    vpos, vnrm, vao = o.vertices()
    t = o.triangles()
    assert np.array_equal(bits(a["vertex_pos"]), bits(vpos)) and np.array_equal(bits(a["vertex_normal"]), bits(vnrm))
    assert np.array_equal(a["vertex_ao"], vao) and np.array_equal(a["tri_index"], t["idx"])
    assert np.array_equal(bits(a["tri_center"]), bits(t["center"])) and np.array_equal(bits(a["tri_normal"]), bits(t["normal"]))
    assert np.array_equal(a["tri_color32"], t["color32"]) and int(a["tri_color32"][0]) == 0xFF0000
    assert float(np.abs(a["vertex_pos"]).max()) == 0.5          # not rescaled to 1.2
    assert not a["tri_d"].any() and not a["tri_e"].any()         # the tail never ran
