"""GPU SAH builder (mi355_build_bvh): the reference's exact tree, byte for byte.

Pins: the `.bvh` hashes of the reference's own scalar builder (tests/golden/reference_pins.json); beyond the three
shipped meshes the ORACLE's builder is the checker (oracle/oracle.cc, itself compared with the reference's own
CreateBVH on small soups in tests/test_refcore_pins.py), on meshes built to hit the order-dependent corners: equal
centroids, signed zeros, flat and tiny nodes.  The host layer's sorted-sweep builder must agree with both."""
import hashlib
import json
import os
import zlib

import numpy as np
import pytest

import renderer_amd as R

pytestmark = pytest.mark.gpu
PINS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_pins.json")))
MESHES = ["dragon_vis.ply", "statue.ply", "chessboard.tri"]


class OracleTree:
    """The checker: the oracle's restatement of BVH.cc:96-371 + Raytracer.cc:651-718 on the same file."""
    def __init__(self, path):
        from oracle import oracle_ctypes as O
        self.s = O.Scene(R.assets.oracle_path(os.path.basename(path)) if path.endswith(".3ds") else path)
        self.s.bvh_build()

    def bvh_arrays(self):
        return self.s.bvh()

    def bvh_info(self):
        return (self.s.num_nodes, self.s.nt, self.s.max_depth)


def both_builders(path):
    d = R.Scene(path)
    d.bvh_create("device")
    assert d.bvh_built_on_device
    h = R.Scene(path)
    h.bvh_create("host")
    assert not h.bvh_built_on_device
    return d, h


def assert_same_tree(d, h, path=None):
    """device builder == oracle builder (the checker) == host builder"""
    trees = [("host", h)] + ([("oracle", OracleTree(path))] if path else [])
    dn, di = d.bvh_arrays()
    for name, t in trees:
        hn, hi = t.bvh_arrays()
        assert dn.shape == hn.shape, "%s: node count %d vs %d" % (name, dn.shape[0], hn.shape[0])
        assert np.array_equal(di, hi), "%s: triangle lists differ" % name
        bad = np.argwhere((dn != hn).any(axis=1))
        assert bad.size == 0, "%s: first differing node %d: %s vs %s" % (name, bad[0, 0], dn[bad[0, 0]], hn[bad[0, 0]])
        assert d.bvh_info()[2] == t.bvh_info()[2]        # max depth


@pytest.mark.parametrize("mesh", MESHES)
def test_gpu_builder_emits_the_reference_cache_bytes(mesh):
    d, h = both_builders(R.assets.mesh_path(mesh))
    assert_same_tree(d, h, R.assets.mesh_path(mesh))
    nodes, idx = d.bvh_arrays()
    pin = PINS["bvh"][mesh]
    assert nodes.shape[0] == pin["nodes"]
    blob = np.array([nodes.shape[0], d.nt], np.uint32).tobytes() + nodes.tobytes() + idx.tobytes()
    sha = hashlib.sha256(blob).hexdigest()
    assert sha.startswith(pin["sha_prefix"]) and sha.endswith(pin["sha_suffix"])


def test_gpu_builder_on_the_3ds_model():
    """legocar.3ds (many small parts, exact-duplicate vertices per face): both builders, same tree."""
    d, h = both_builders(R.assets.mesh_path("legocar.3ds"))
    assert_same_tree(d, h, R.assets.mesh_path("legocar.3ds"))


def write_ply(path, verts, faces):
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nelement face %d\nend_header\n" % (len(verts), len(faces)))
        for v in verts:
            f.write("%s %s %s 60\n" % tuple(repr(float(x)) for x in v))
        for t in faces:
            f.write("3 %d %d %d\n" % tuple(t))


def soup(rng, n_tri, snap=None, scale=1.0):
    """n_tri small random triangles; snap = grid step to force equal coordinates / centroids"""
    c = rng.uniform(-1, 1, (n_tri, 1, 3)) * scale
    v = c + rng.uniform(-0.08, 0.08, (n_tri, 3, 3)) * scale
    if snap:
        v = np.round(v / snap) * snap
    verts = v.reshape(-1, 3)
    faces = np.arange(3 * n_tri).reshape(n_tri, 3)
    return verts, faces


CASES = {
    "random_3000": lambda rng: soup(rng, 3000),
    "snapped_grid_2000 (many equal centroids and box faces)": lambda rng: soup(rng, 2000, snap=0.125),
    "coarse_grid_600 (ties everywhere)": lambda rng: soup(rng, 600, snap=0.5),
    "three_triangles (single leaf)": lambda rng: soup(rng, 3),
    "four_triangles": lambda rng: soup(rng, 4),
    "five_triangles": lambda rng: soup(rng, 5),
    "seventeen_triangles": lambda rng: soup(rng, 17),
    "flat_in_z_500 (one axis below the 1e-4 extent)": lambda rng: (lambda vf: (vf[0] * np.array([1, 1, 0.0]), vf[1]))(soup(rng, 500)),
    "duplicates_300 (identical triangles cannot be separated)": lambda rng: (lambda vf: (np.tile(vf[0], (10, 1)), np.arange(900).reshape(300, 3)))(soup(rng, 30)),
    "long_thin_1000": lambda rng: (lambda vf: (vf[0] * np.array([40.0, 1, 0.2]), vf[1]))(soup(rng, 1000)),
    # more than 4096 triangles in a node: the first levels are split by many workgroups (k_big_bin / _eval / _scatter)
    "random_20000": lambda rng: soup(rng, 20000),
    "snapped_grid_30000 (chunked nodes with equal centroids)": lambda rng: soup(rng, 30000, snap=0.0625),
    "exactly_4096": lambda rng: soup(rng, 4096),
    "chunk_plus_one_4097": lambda rng: soup(rng, 4097),
    "two_chunks_8192": lambda rng: soup(rng, 8192),
    "duplicates_9000 (a chunked node that cannot be split)": lambda rng: (lambda vf: (np.tile(vf[0], (3000, 1)), np.arange(27000).reshape(9000, 3)))(soup(rng, 3)),
    "flat_in_z_12000": lambda rng: (lambda vf: (vf[0] * np.array([1, 1, 0.0]), vf[1]))(soup(rng, 12000)),
}


@pytest.mark.parametrize("case", list(CASES), ids=[k.split(" ")[0] for k in CASES])
def test_gpu_builder_matches_host_builder_on_synthetic_meshes(case, tmp_path):
    rng = np.random.default_rng(zlib.crc32(case.encode()))         # fixed inputs from run to run
    verts, faces = CASES[case](rng)
    p = str(tmp_path / "m.ply")
    write_ply(p, verts, faces)
    d, h = both_builders(p)
    assert_same_tree(d, h, p)


def test_gpu_builder_keeps_the_sign_of_zero_the_reference_keeps(tmp_path):
    """Boxes are accumulated with std::min/std::max in list order, so among equal extremes the FIRST wins -- which is
    visible as the sign of a zero coordinate in the stored boxes.  Symmetric coordinates around 0 with both +0 and -0."""
    rng = np.random.default_rng(7)
    n = 400
    base = rng.choice(np.array([-1.0, -0.5, -0.0, 0.0, 0.5, 1.0]), (n, 3, 3))
    base += rng.choice(np.array([0.0, -0.0]), (n, 3, 3))
    # make sure the bounding box stays symmetric so that the loader's re-centring keeps the zeros
    verts = np.concatenate([base.reshape(-1, 3), [[-1, -1, -1], [1, 1, 1], [1, -1, 1]]])
    faces = np.concatenate([np.arange(3 * n).reshape(n, 3), [[3 * n, 3 * n + 1, 3 * n + 2]]])
    p = str(tmp_path / "z.ply")
    write_ply(p, verts, faces)
    d, h = both_builders(p)
    dn, _ = d.bvh_arrays()
    boxes = dn[:, :6]
    assert ((boxes == 0x80000000).any() or (boxes == 0).any()), "the case is meant to produce zero box coordinates"
    assert_same_tree(d, h, p)


def test_signed_zeros_in_chunked_nodes(tmp_path):
    """the same with nodes large enough to be split by several workgroups: the first zero in list order may sit in any chunk"""
    rng = np.random.default_rng(11)
    n = 15000
    base = rng.choice(np.array([-1.0, -0.5, -0.25, -0.0, 0.0, 0.25, 0.5, 1.0]), (n, 3, 3))
    base += rng.choice(np.array([0.0, -0.0]), (n, 3, 3))
    verts = np.concatenate([base.reshape(-1, 3), [[-1, -1, -1], [1, 1, 1], [1, -1, 1]]])
    faces = np.concatenate([np.arange(3 * n).reshape(n, 3), [[3 * n, 3 * n + 1, 3 * n + 2]]])
    p = str(tmp_path / "z.ply")
    write_ply(p, verts, faces)
    d, h = both_builders(p)
    assert_same_tree(d, h, p)


@pytest.mark.parametrize("mesh", MESHES + ["synthetic"])
def test_device_made_traversal_streams_equal_the_host_made_ones(mesh, tmp_path):
    """mi355_build_bvh numbers the nodes and writes the walk / wide / edge / shading records with kernels; mi355_scene_set_bvh
    makes them on the host from the node array (the path of a `.bvh` cache or a foreign tree).  Same bytes, same scalars."""
    if mesh == "synthetic":
        verts, faces = soup(np.random.default_rng(5), 6000)
        path = str(tmp_path / "m.ply")
        write_ply(path, verts, faces)
    else:
        path = R.assets.mesh_path(mesh)
    s = R.Scene(path)
    nodes, idx, _ = s.build_bvh_device()
    dev = s.traversal_state()
    s.set_bvh_arrays(nodes, idx)
    host = s.traversal_state()
    for name, a, b in zip(("scalars", "walk", "edge", "shade"), dev, host):
        assert a.shape == b.shape, name
        bad = np.flatnonzero(a != b)
        assert bad.size == 0, "%s: %d words differ, first at %d: %#x vs %#x" % (name, bad.size, bad[0], a[bad[0]], b[bad[0]])


def test_frames_after_a_device_build_match_the_pins():
    """The tree mi355_build_bvh installs in the context is the one the frames are traced with."""
    s = R.Scene(R.assets.mesh_path("dragon_vis.ply"))
    s.bvh_create("device")
    pin = [p for p in PINS["frames"] if p.get("mesh", "").startswith("dragon") and p.get("mode") == 9]
    cam, lights, n = R.benchmark_frame(0)
    img = s.render(9, cam, lights, n, R.default_opts(640, 360))[0]
    h = R.Scene(R.assets.mesh_path("dragon_vis.ply"))
    h.bvh_create("host")
    assert (img == h.render(9, cam, lights, n, R.default_opts(640, 360))[0]).all()


# ---- traced frames over synthetic trees: the tie rules of the ordered walk ---------------------------------------------
def write_coloured_ply(path, verts, faces, colours):
    with open(path, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nelement face %d\nend_header\n" % (len(verts), len(faces)))
        for v in verts:
            f.write("%s %s %s 200\n" % tuple(repr(float(x)) for x in v))
        for t, c in zip(faces, colours):
            f.write("3 %d %d %d %d %d %d\n" % (tuple(t) + tuple(c)))


@pytest.mark.parametrize("case", ["stacked_duplicates", "coplanar_overlaps", "random_soup"])
def test_traced_frames_on_synthetic_meshes_match_the_oracle(case, tmp_path, oracle):
    """Equal hit distances are where an order-free walk could differ from the reference's first-found-wins: identical
    triangles in different colours (every ray has several hits at exactly the same distance) and overlapping coplanar
    ones.  The production kernel (near-first walk, parts of a ray's walk done by other lanes of its wave) must give the oracle's pixels."""
    rng = np.random.default_rng({"stacked_duplicates": 11, "coplanar_overlaps": 12, "random_soup": 13}[case])
    if case == "stacked_duplicates":
        v0, f0 = soup(rng, 60, scale=1.0)
        verts = np.tile(v0 * 4.0, (5, 1))                                   # five copies of every triangle
        faces = np.arange(verts.shape[0]).reshape(-1, 3)
    elif case == "coplanar_overlaps":
        n = 250
        c = rng.uniform(-1, 1, (n, 1, 3)) * np.array([1, 1, 0.0])
        tri = c + rng.uniform(-0.4, 0.4, (n, 3, 3)) * np.array([1, 1, 0.0])
        tri[:, :, 2] = np.round(rng.uniform(-1, 1, (n, 1)) * 2) / 2          # a few shared planes z = -1, -0.5, .. 1
        verts = tri.reshape(-1, 3)
        faces = np.arange(3 * n).reshape(n, 3)
    else:
        verts, faces = soup(rng, 1500)
        verts = verts * np.array([1.0, 1.0, 1.0])
    colours = rng.integers(30, 255, (faces.shape[0], 3))
    p = str(tmp_path / (case + ".ply"))
    write_coloured_ply(p, verts, faces, colours)
    g = R.Scene(p)
    g.bvh_create("device")
    o = oracle.Scene(p)
    o.bvh_build()
    assert g.walk_info()[0] == 1
    for frame in (0, 23, 61):
        cam, lights, n = R.benchmark_frame(frame)
        ocam, olights, on = oracle.benchmark_frame(frame)
        W, H = 400, 300
        img, f32, st = g.render(9, cam, lights, n, R.default_opts(W, H), want_f32=True)
        oimg, of32, ost = o.render(9, ocam, olights, on, oracle.default_opts(W, H, threads=os.cpu_count() or 1), want_f32=True)
        assert int((img != oimg).sum()) == 0, "%s frame %d: %d pixels differ" % (case, frame, int((img != oimg).sum()))
        assert float(np.abs(f32 - of32).max()) == 0.0
        assert (st.normal_rays, st.shadow_rays) == (ost.normal_rays, ost.shadow_rays)
        assert int((img != 0).sum()) > 500, "the case is meant to put the mesh on screen"
        if frame == 23:
            # a closest-hit ray walked by several lanes at once (its hit = the atomic minimum of what they find: nearest,
            # then lowest triangle): at every hand-over threshold and in every register build the same frame, bit for bit
            for knobs in (dict(sharemin=1), dict(sharemin=1, bpc=3), dict(sharemin=1, bpc=4), dict(sharemin=4, bpc=2), dict(sharemin=40),
                          dict(noshare=1), dict(sharemin=1, exact=1)):
                img2, f2, st2 = g.render(9, cam, lights, n, R.default_opts(W, H, tune=R.tune(**knobs)), want_f32=True)
                assert np.array_equal(img2, oimg) and np.array_equal(f2, of32), knobs
                assert (st2.normal_rays, st2.shadow_rays) == (ost.normal_rays, ost.shadow_rays), knobs


# ---- deep trees: the kernel's LDS (colour rows + one stack row per level + the sharing rows) is sized at launch ------------------
def chain_tree(vpos, tri_index):
    """A legal tree as deep as a tree of T triangles can be: inner node k = (leaf of triangle k, inner node k + 1), boxes the exact
    unions; the reference's node format in its pre-order (Raytracer.cc:651-718: inner: idxLeft, idxRight; leaf: count | 1 << 31, start)."""
    T = tri_index.shape[0]
    tv = vpos[tri_index]                                   # (T, 3, 3)
    lo, hi = tv.min(axis=1).astype(np.float32), tv.max(axis=1).astype(np.float32)
    slo, shi = lo.copy(), hi.copy()
    for k in range(T - 2, -1, -1):                         # suffix unions
        slo[k] = np.minimum(lo[k], slo[k + 1]); shi[k] = np.maximum(hi[k], shi[k + 1])
    nodes = np.zeros((2 * T - 1, 8), np.uint32)
    f = nodes[:, :6].view(np.float32)
    for k in range(T - 1):
        i = 2 * k                                          # inner k, its leaf at i + 1, the rest at i + 2
        f[i, :3], f[i, 3:] = slo[k], shi[k]
        nodes[i, 6], nodes[i, 7] = i + 1, i + 2
        f[i + 1, :3], f[i + 1, 3:] = lo[k], hi[k]
        nodes[i + 1, 6], nodes[i + 1, 7] = 0x80000001, k
    i = 2 * (T - 1)
    f[i, :3], f[i, 3:] = lo[T - 1], hi[T - 1]
    nodes[i, 6], nodes[i, 7] = 0x80000001, T - 1
    return nodes, np.arange(T, dtype=np.int32)


@pytest.mark.parametrize("n_tri", [31, 47])
def test_deep_trees_render_in_every_build(n_tri, tmp_path, oracle):
    """The ordered walk keeps a stack row per tree level in LDS, beside the colour rows of the ray tree's levels and the rows of
    the work sharing; all of it is asked for at launch.  The deepest trees the ordered walk accepts (MI_MAX_STACK = 48 levels), the
    deepest ray trees (4) and 4 spp together must still launch -- in whatever build fits -- and give the oracle's frame over the
    same tree."""
    rng = np.random.default_rng(100 + n_tri)
    tri = rng.uniform(-0.7, 0.7, (n_tri, 1, 3)) + rng.uniform(-0.3, 0.3, (n_tri, 3, 3))       # large enough to fill the picture
    verts, faces = tri.reshape(-1, 3), np.arange(3 * n_tri).reshape(n_tri, 3)
    p = str(tmp_path / "deep.ply")
    write_coloured_ply(p, verts, faces, rng.integers(40, 255, (n_tri, 3)))
    g = R.Scene(p)
    a = g.arrays()
    nodes, idx = chain_tree(np.array(a["vertex_pos"]), np.array(a["tri_index"]))
    g.set_bvh_arrays(nodes, idx)
    ok, depth, n_nodes, _ = g.walk_info()
    assert ok == 1 and depth == n_tri and n_nodes == 2 * n_tri - 1        # (inner levels n_tri - 1, one stack entry more)
    bvh = str(tmp_path / "deep.bvh")
    with open(bvh, "wb") as fp:
        fp.write(np.array([nodes.shape[0], idx.shape[0]], np.uint32).tobytes()); fp.write(nodes.tobytes()); fp.write(idx.tobytes())
    o = oracle.Scene(p)
    assert o.bvh_load(bvh) == nodes.shape[0]
    cam, lights, n = R.benchmark_frame(7)
    ocam, olights, on = oracle.benchmark_frame(7)
    W, H = 320, 200
    for mode, depth_rays in ((9, 3), (9, 4), (10, 4)):
        want, wf, wst = o.render(mode, ocam, olights, on, oracle.default_opts(W, H, max_ray_depth=depth_rays, threads=os.cpu_count() or 1), want_f32=True)
        assert int((want != 0).sum()) > 3000
        for knobs in (dict(), dict(bpc=2), dict(bpc=3), dict(bpc=4), dict(noshare=1), dict(sharemin=1, bpc=4), dict(exact=1), dict(reforder=1)):
            img, f32, st = g.render(mode, cam, lights, n, R.default_opts(W, H, max_ray_depth=depth_rays, tune=R.tune(**knobs)), want_f32=True)
            assert np.array_equal(img, want) and np.array_equal(f32, wf), (mode, depth_rays, knobs)
            assert (st.normal_rays, st.shadow_rays) == (wst.normal_rays, wst.shadow_rays), (mode, depth_rays, knobs)
    # batches of frames (their own kernel variant), the deepest ray trees
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    frames = [R.benchmark_frame(k) for k in (7, 8, 9, 10)]
    for knobs in (dict(), dict(bpc=3), dict(bpc=4)):
        bo = R.default_opts(W, H, max_ray_depth=4, tune=R.tune(**knobs))
        bufs = [torch.zeros((H, W), dtype=torch.int32, device=dev) for _ in frames]
        g.render_batch_device(9, [f[0] for f in frames], [f[1] for f in frames], frames[0][2], bo, [b.data_ptr() for b in bufs], W * 4, None,
                              torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize(dev)
        for k, f in enumerate(frames):
            assert np.array_equal(bufs[k].cpu().numpy().view(np.uint32), g.render(9, f[0], f[1], f[2], R.default_opts(W, H, max_ray_depth=4))[0]), (knobs, k)
