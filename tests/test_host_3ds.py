"""The host layer's from-scratch .3ds reader (renderer_amd/csrc/host/load_3ds.cc) against the REAL lib3ds 1.3.0:
bit for bit against the committed dump of the reference's legocar.3ds (made by oracle/ref3ds/dump3ds.c with the real
library, scripts/make_3ds_golden.sh) and, where the reference tree is present, against the real library run on
generated files (smoothing groups, shared and duplicated points, every colour chunk variant, unsorted, duplicate and
missing material names)."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

import renderer_amd as R

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB3DS = "/root/reference/lib3ds-1.3.0/lib3ds"


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


# ---- a minimal .3ds writer (test data only) ----------------------------------------------------------
def chunk(cid, payload=b"", *children):
    body = payload + b"".join(children)
    return struct.pack("<HI", cid, 6 + len(body)) + body


def cstr(s):
    return s.encode() + b"\0"


def color(kind, rgb):
    if kind == "24":
        return chunk(0x0011, bytes(rgb))
    if kind == "lin24":
        return chunk(0x0012, bytes(rgb))
    if kind == "f":
        return chunk(0x0010, struct.pack("<3f", *[c / 255.0 for c in rgb]))
    return chunk(0x0013, struct.pack("<3f", *[c / 255.0 for c in rgb]))


def material(name, kinds, rgbs, two_sided):
    kids = [chunk(0xA000, cstr(name)), chunk(0xA020, b"", *[color(k, c) for k, c in zip(kinds, rgbs)])]
    if two_sided:
        kids.append(chunk(0xA081))
    return chunk(0xAFFF, b"", *kids)


def mesh(name, points, faces, smoothing, groups, matrix=None):
    pa = chunk(0x4110, struct.pack("<H", len(points)) + np.asarray(points, "<f4").tobytes())
    fbody = struct.pack("<H", len(faces)) + b"".join(struct.pack("<4H", a, b, c, 7) for a, b, c in faces)
    kids = []
    if smoothing is not None:
        kids.append(chunk(0x4150, np.asarray(smoothing, "<u4").tobytes()))
    for mname, idx in groups:
        kids.append(chunk(0x4130, cstr(mname) + struct.pack("<H", len(idx)) + np.asarray(idx, "<u2").tobytes()))
    fa = chunk(0x4120, fbody, *kids)
    parts = [pa, fa]
    if matrix is not None:
        parts.append(chunk(0x4160, np.asarray(matrix, "<f4").tobytes()))
    return chunk(0x4000, cstr(name), chunk(0x4100, b"", *parts))


def random_file(seed):
    rng = np.random.default_rng(seed)
    mats, names = [], []
    for m in range(int(rng.integers(1, 5))):
        name = ["red", "Blue", "a mat", "zinc", "red"][int(rng.integers(0, 5))]
        kinds = [["24"], ["lin24"], ["f"], ["linf"], ["24", "lin24"], ["lin24", "24"], ["f", "linf"], ["linf", "f", "24"]][int(rng.integers(0, 8))]
        rgbs = [tuple(int(v) for v in rng.integers(0, 256, 3)) for _ in kinds]
        mats.append(material(name, kinds, rgbs, bool(rng.integers(0, 2))))
        names.append(name)
    objs = []
    for o in range(int(rng.integers(1, 5))):
        nP = int(rng.integers(3, 40))
        pts = rng.uniform(-5, 5, (nP, 3)).astype(np.float32)
        if rng.random() < 0.5:
            pts = np.round(pts)                                    # coplanar faces -> equal normals (the 1e-5 dedup)
        if rng.random() < 0.3:
            pts[nP // 2] = pts[0]                                  # coincident points
        nF = int(rng.integers(1, 60))
        faces = rng.integers(0, nP, (nF, 3))
        if rng.random() < 0.2:
            faces[0] = [0, 0, 1]                                   # degenerate: zero normal -> (1,0,0) rule
        smoothing = None if rng.random() < 0.2 else rng.choice([0, 1, 2, 3, 4, 0xffffffff], nF)
        groups = []
        for g in range(int(rng.integers(0, 3))):
            gname = (names + ["nobody"])[int(rng.integers(0, len(names) + 1))]
            groups.append((gname, rng.choice(nF, int(rng.integers(1, nF + 1)), replace=False)))
        matrix = None
        if rng.random() < 0.5:
            matrix = np.concatenate([np.eye(3) * rng.uniform(0.5, 2), rng.uniform(-1, 1, (1, 3))]).astype(np.float32)
        oname = ["wheel", "Body", "axle 2", "body", "wheel"][int(rng.integers(0, 5))]
        objs.append(mesh(oname, pts, faces, smoothing, groups, matrix))
    order = list(mats) + list(objs)
    if rng.random() < 0.5:
        rng.shuffle(order)                                          # materials may come after the objects that name them
    mdata = chunk(0x3D3D, b"", chunk(0x3D3E, struct.pack("<I", 3)), *order)
    return chunk(0x4D4D, b"", chunk(0x0002, struct.pack("<I", 3)), mdata)


@pytest.fixture(scope="module")
def real_lib3ds():
    """oracle/_ref/dump3ds: the real library, buildable only where the reference tree is."""
    exe = os.path.join(REPO, "oracle", "_ref", "dump3ds")
    if os.path.isdir(LIB3DS) and shutil.which("gcc"):
        subprocess.run(["make", "-C", os.path.join(REPO, "oracle", "ref3ds")], check=True, capture_output=True)
    if not os.path.exists(exe):
        pytest.skip("the reference's lib3ds is not here")
    return exe


def test_legocar_matches_the_real_lib3ds_dump(tmp_path):
    out = str(tmp_path / "mine.r3ds")
    R.dump_3ds(R.assets.mesh_path("legocar.3ds"), out)
    want = open(R.assets.oracle_path("legocar.3ds"), "rb").read()
    got = open(out, "rb").read()
    assert struct.unpack("<I", got[4:8])[0] == 10992
    assert got == want


@pytest.mark.parametrize("seed", range(40))
def test_generated_files_match_the_real_lib3ds(real_lib3ds, tmp_path, seed):
    src = str(tmp_path / "g.3ds")
    open(src, "wb").write(random_file(seed))
    ref = str(tmp_path / "ref.r3ds")
    p = subprocess.run([real_lib3ds, src, ref], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    R.dump_3ds(src, str(tmp_path / "mine.r3ds"))
    assert open(str(tmp_path / "mine.r3ds"), "rb").read() == open(ref, "rb").read()


def test_scene_from_3ds_equals_the_oracle_scene(oracle):
    """After the loader's common tail (centre, rescale, triangle precompute) the host scene read from the .3ds file
    equals the oracle's scene built from the real lib3ds dump."""
    h = R.Scene(R.assets.mesh_path("legocar.3ds"))
    o = oracle.Scene(R.assets.oracle_path("legocar.3ds"))
    assert (h.nv, h.nt) == (3 * 10992, 10992)
    vpos, vnrm, vao = o.vertices()
    t = o.triangles()
    a = h.arrays()
    assert np.array_equal(bits(a["vertex_pos"]), bits(vpos))
    assert np.array_equal(bits(a["vertex_normal"]), bits(vnrm))
    assert np.array_equal(a["vertex_ao"], vao) and set(np.unique(vao)) == {60}
    assert np.array_equal(a["tri_index"], t["idx"])
    for k in ("center", "normal", "colorf"):
        assert np.array_equal(bits(a["tri_" + k]), bits(t[k])), k
    assert np.array_equal(bits(a["tri_d"]), bits(t["plane"][:, 0:4]))
    assert np.array_equal(bits(a["tri_e"]), bits(t["plane"][:, 4:13]))
    assert np.array_equal(a["tri_color32"], t["color32"])
    assert np.array_equal(a["tri_two_sided"], t["two_sided"])
    assert int(t["two_sided"].sum()) == 120                     # legocar's two-sided material: the twoSided paths see real data


def test_bad_3ds_files_are_refused(tmp_path):
    good = random_file(3)
    for name, data, msg in (("trunc.3ds", good[: len(good) // 2], "Malformed"),
                            ("magic.3ds", b"\x11\x22" + good[2:], "couldn't load"),
                            ("empty.3ds", chunk(0x4D4D, b"", chunk(0x3D3D)), "no meshes")):
        p = str(tmp_path / name)
        open(p, "wb").write(data)
        with pytest.raises(R.Mi355Error, match=msg):
            R.Scene(p)
    mirrored = chunk(0x4D4D, b"", chunk(0x3D3D, b"", mesh("m", [[0, 0, 0], [1, 0, 0], [0, 1, 0]], [[0, 1, 2]], [1], [],
                                                          np.array([[-1, 0, 0], [0, 1, 0], [0, 0, 1], [0, 0, 0]], np.float32))))
    p = str(tmp_path / "mirror.3ds")
    open(p, "wb").write(mirrored)
    with pytest.raises(R.Mi355Error, match="mirrored"):
        R.Scene(p)
