"""Differential checks of the host layer's loaders against the oracle's (which read `.ply` lines with the same
`std::istringstream >>` extraction the reference uses, Loader.cc:354-409): generated files with odd but legal tokens
(signs, exponents, tabs, CRLF), tokens the stream extraction rejects ("inf", "1e", "--1", out-of-range numbers, text), short
and long lines -- both loaders must agree on every array or both must refuse the file.  Same for `.tri` files with every
magic, truncations and bad indices."""
import os
import struct

import numpy as np
import pytest

import renderer_amd as R

TOK_F = ["1", "-2.5", "+3", "1e2", "1E-3", ".5", "5.", "1e", "e5", "0x10", "inf", "nan", "-inf", "1.5abc", "abc", "1,5", "--1",
         "1e+", "007", "1e400", "1e-400", "-0", "", " ", "\t", "3.4028236e38", "1.17549435e-38", "1e-46", "+.5e+1", "1.2.3",
         "1..2", "1e2e3", "  7"]
TOK_U = ["0", "1", "2", "255", "256", "-1", "+4", "4294967295", "4294967296", "99999999999999999999", "1.5", "abc", "0x5", "",
         "12abc", "-0", "007"]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def both(path, oracle):
    herr = oerr = h = o = None
    try:
        h = R.Scene(path)
    except R.Mi355Error as e:
        herr = str(e)
    try:
        o = oracle.Scene(path)
    except RuntimeError as e:
        oerr = str(e)
    return h, o, herr, oerr


def same_scene(h, o):
    a = h.arrays()
    vpos, vnrm, vao = o.vertices()
    t = o.triangles()
    return ((h.nv, h.nt) == (o.nv, o.nt) and np.array_equal(bits(a["vertex_pos"]), bits(vpos))
            and np.array_equal(bits(a["vertex_normal"]), bits(vnrm)) and np.array_equal(a["vertex_ao"], vao)
            and np.array_equal(a["tri_index"], t["idx"]) and np.array_equal(a["tri_color32"], t["color32"])
            and np.array_equal(bits(a["tri_colorf"]), bits(t["colorf"])))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_ply_lines_read_like_the_stream_extraction(oracle, tmp_path, seed):
    rng = np.random.default_rng(seed)
    for it in range(80):
        nv, nf = int(rng.integers(3, 8)), int(rng.integers(1, 5))
        head = ["ply", "format ascii 1.0", ["element vertex %d", "element vertex   %d", "element vertexes %d"][int(rng.integers(0, 3))] % nv,
                "property float x", "element face %d" % nf, "end_header"]
        lines = []
        for i in range(nv):
            if rng.random() < 0.5:
                lines.append("%g %g %g %d" % (rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(-1, 1), rng.integers(0, 256)))
            else:
                sep = [" ", "  ", "\t", " \t "][int(rng.integers(0, 4))]
                lines.append(sep.join([TOK_F[int(rng.integers(0, len(TOK_F)))] for _ in range(3)] + [TOK_U[int(rng.integers(0, len(TOK_U)))]])
                             + ("\r" if rng.random() < 0.1 else ""))
        for i in range(nf):
            if rng.random() < 0.5:
                lines.append("3 %d %d %d %d %d %d" % (*rng.integers(0, nv, 3), *rng.integers(0, 256, 3)))
            else:
                lines.append(" ".join([TOK_U[int(rng.integers(0, len(TOK_U)))] if rng.random() < 0.4 else str(int(rng.integers(0, nv)))
                                       for _ in range(int(rng.integers(1, 8)))]))
        p = str(tmp_path / ("f%d.ply" % it))
        open(p, "w").write("\n".join(head + lines) + "\n")
        h, o, herr, oerr = both(p, oracle)
        assert (herr is None) == (oerr is None), "file %d: host %r, oracle %r\n%s" % (it, herr, oerr, "\n".join(lines))
        if herr is None:
            assert same_scene(h, o), "file %d:\n%s" % (it, "\n".join(lines))


@pytest.mark.parametrize("seed", [1, 2])
def test_tri_files_with_every_magic_truncations_and_bad_indices(oracle, tmp_path, seed):
    rng = np.random.default_rng(seed)
    for it in range(60):
        magic = [0xDEADBEEF, 0xDEADC0DE, None, 0x12345678][int(rng.integers(0, 4))]
        with_n, with_c = magic == 0xDEADC0DE, magic in (0xDEADBEEF, 0xDEADC0DE)
        blob = b"" if magic is None else struct.pack("<I", magic)
        total = 0
        for blk in range(int(rng.integers(1, 4))):
            nv, nt = int(rng.integers(3, 9)), int(rng.integers(1, 6))
            blob += struct.pack("<I", nv) + rng.uniform(-2, 2, (nv, 6 if with_n else 3)).astype("<f4").tobytes()
            blob += struct.pack("<I", nt)
            for t in range(nt):
                idx = rng.integers(0, total + nv + (2 if rng.random() < 0.05 else 0), 3)
                blob += struct.pack("<3I", *[int(x) for x in idx])
                if with_c:
                    blob += rng.choice([0.0, 0.5, 1.0, 0.999, 1.7, -0.2], 3).astype("<f4").tobytes()
            total += nv
        if rng.random() < 0.3:
            blob = blob[: int(rng.integers(1, len(blob)))]
        p = str(tmp_path / ("t%d.tri" % it))
        open(p, "wb").write(blob)
        h, o, herr, oerr = both(p, oracle)
        assert (herr is None) == (oerr is None), "file %d (magic %r): host %r, oracle %r" % (it, magic, herr, oerr)
        if herr is None:
            assert same_scene(h, o), "file %d (magic %r)" % (it, magic)


@pytest.mark.parametrize("seed", [1, 2])
def test_host_bvh_builder_equals_the_oracle_builder_on_random_soups(oracle, tmp_path, seed):
    """The host layer's sorted-sweep builder against the oracle's restatement of BVH.cc (plane by plane): same `.bvh`
    bytes on soups with ties, duplicates, flat and thin axes (the device builder is compared with the host builder in
    tests/test_gpu_bvh.py and scripts/fuzz_parity.py)."""
    for it in range(40):
        rng = np.random.default_rng(seed * 1000003 + it)
        n_tri = int(rng.choice([1, 2, 3, 4, 5, 8, 9, 30, 200, 1500]))
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
        h = R.Scene(p)
        if not np.isfinite(h.arrays()["vertex_pos"]).all():
            continue                                   # collapsed to a point by the snapping: the rescale makes NaNs of it
        h.bvh_create("host")
        o = oracle.Scene(p)
        o.bvh_build()
        o.bvh_save(p + ".obvh")
        hn, hi = h.bvh_arrays()
        blob = np.array([hn.shape[0], h.nt], np.uint32).tobytes() + hn.tobytes() + hi.tobytes()
        assert blob == open(p + ".obvh", "rb").read(), "soup %d (%d triangles, snap %s, stretch %s)" % (it, len(verts) // 3, snap, stretch)


@pytest.mark.parametrize("flip", [False, True])
def test_ra2_files(oracle, tmp_path, monkeypatch, flip):
    """`.ra2` (Loader.cc:224-275): raw 36-byte triangles, vertices stored (y, z, x), a trailing partial triangle ignored,
    `$RA2` flips the winding."""
    if flip:
        monkeypatch.setenv("RA2", "1")
    else:
        monkeypatch.delenv("RA2", raising=False)
    rng = np.random.default_rng(7)
    for it, extra in enumerate((0, 0, 5, 35)):
        n = int(rng.integers(1, 40))
        blob = rng.uniform(-3, 3, (n, 9)).astype("<f4").tobytes() + bytes(extra)
        p = str(tmp_path / ("m%d.ra2" % it))
        open(p, "wb").write(blob)
        h, o, herr, oerr = both(p, oracle)
        assert herr is None and oerr is None
        assert (h.nv, h.nt) == (3 * n, n) and same_scene(h, o)
        t = o.triangles()
        a = h.arrays()
        assert np.array_equal(bits(a["tri_normal"]), bits(t["normal"])) and np.array_equal(bits(a["tri_e"]), bits(t["plane"][:, 4:13]))
