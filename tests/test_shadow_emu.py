"""CPU validation of the shadow map's tile kernels without a GPU (rows b8 of SURVEY 8): renderer_amd/csrc/sm_core.h -- what a
thread of k_sm_prep computes for its triangle (projection, rows, the conservative column range, the per-row edge steps, the band or
coarse-band lists it is entered in) and what a thread of k_sm_tiles plots for a (triangle, row) inside its tile's columns (edge walkers
brought to the row with ff_add, the span entered at the tile's first column, bisection when the estimate drifted) -- is compiled for the
host (tests/emu/emu_shadow.hip) and driven tile by tile against the oracle's serial Light.cc:84-296.  Bit for bit, every texel."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import renderer_amd as R
from emu import emu


@pytest.fixture(scope="module")
def scenes(oracle):
    cache = {}

    def get(mesh):
        if mesh not in cache:
            p = R.assets.mesh_path(mesh)
            hs = R.Scene(p)
            cache[mesh] = (hs, oracle.Scene(p), emu.scene_streams(hs))
        return cache[mesh]
    return get


def both(oracle, hs, osc, streams, pos, size):
    cam, _, _ = R.benchmark_frame(0)
    ocam, _, _ = oracle.benchmark_frame(0)
    p32 = np.array(pos, np.float32)
    m, st = emu.shadowmap(hs, R.light(p32, cam), size, streams)
    om = osc.shadowmap(oracle.light(p32, ocam), size=size)
    return m, om, st


@pytest.mark.parametrize("mesh", ["chessboard.tri", "dragon_vis.ply", "statue.ply"])
@pytest.mark.parametrize("size", [1024, 37, 600, 1025, 2050])
def test_the_tile_kernels_arithmetic_draws_the_oracles_map(oracle, scenes, mesh, size):
    hs, osc, streams = scenes(mesh)
    for pos in ((3.394, 3.394, 4.8), (-2.0, 3.5, 3.0)):
        m, om, st = both(oracle, hs, osc, streams, pos, size)
        assert np.array_equal(m, om), "%s %d %s: %d texels differ" % (mesh, size, pos, int((m != om).sum()))
        assert st["drawn"] > 1000 and st["pixels"] >= int((om > -1e30).sum())


@pytest.mark.parametrize("mesh,size,pos", [
    ("chessboard.tri", 1024, (0.9, 0.8, 0.7)),          # a light close to the board: triangles many times the map's size, most of them cut
    ("chessboard.tri", 1024, (0.05, 0.02, 0.3)),        # ... above the middle of it: geometry on every side, degenerate projections
    ("chessboard.tri", 333, (0.4, -0.3, 0.12)),         # ... a hand above the board, off centre
    ("dragon_vis.ply", 777, (1.2, -0.9, 0.8)),
    ("dragon_vis.ply", 512, (0.5, 0.45, 0.5)),          # at the edge of the dragon's box (inside it the ORACLE's serial spans take minutes)
], ids=lambda v: str(v).replace(" ", ""))
def test_lights_close_to_and_inside_the_meshes(oracle, scenes, mesh, size, pos):
    """Spans that start far left of the map or of a tile (the exact skip-ahead and its bisection), triangles behind and through the
    light's plane, triangles whose column range falls back to "every column"."""
    hs, osc, streams = scenes(mesh)
    m, om, st = both(oracle, hs, osc, streams, pos, size)
    assert np.array_equal(m, om), "%d texels differ" % int((m != om).sum())


def soup(rng):
    n_tri = int(rng.choice([1, 7, 60, 400, 3000]))
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
    return v.reshape(-1, 3).astype(np.float32)


@pytest.mark.parametrize("seed", range(30))
def test_random_soups(oracle, tmp_path, seed):
    """Tiny to huge triangles, slivers, exact duplicates, snapped grids, flat clouds; lights outside, at the edge of and inside the
    cloud; map sizes that are and are not multiples of the tiles and bands."""
    rng = np.random.default_rng(5000 + seed)
    verts = soup(rng)
    p = str(tmp_path / ("soup%d.ply" % seed))
    with open(p, "w") as f:
        f.write("ply\nformat ascii 1.0\nelement vertex %d\nelement face %d\nend_header\n" % (len(verts), len(verts) // 3))
        for q in verts:
            f.write("%r %r %r 60\n" % (float(q[0]), float(q[1]), float(q[2])))
        for t in range(len(verts) // 3):
            f.write("3 %d %d %d\n" % (3 * t, 3 * t + 1, 3 * t + 2))
    try:
        hs = R.Scene(p)
    except R.Mi355Error:
        pytest.skip("degenerate soup: the loader declines it")
    osc = oracle.Scene(p)
    streams = emu.scene_streams(hs)
    for trial in range(3):
        lp = rng.uniform(-1, 1, 3) * float(rng.choice([0.3, 1.2, 4.0]))
        msize = int(rng.choice([33, 64, 257, 512, 1024, 1500]))
        m, om, st = both(oracle, hs, osc, streams, lp, msize)
        assert np.array_equal(m, om), "seed %d trial %d: %d triangles, light %s, map %d: %d texels differ" % (
            seed, trial, len(verts) // 3, lp.tolist(), msize, int((m != om).sum()))


@pytest.mark.parametrize("size", [4097, 8192])
def test_the_largest_maps_the_tile_kernels_take(oracle, scenes, size):
    """8192 rows = SMT_BANDS bands of SMT_H rows: the list numbering (bands, an empty one, coarse bands, an empty one) at its limit."""
    hs, osc, streams = scenes("chessboard.tri")
    m, om, st = both(oracle, hs, osc, streams, (3.394, 3.394, 4.8), size)
    assert np.array_equal(m, om), "%d texels differ" % int((m != om).sum())


def test_a_triangle_is_entered_in_bands_or_in_coarse_bands(oracle, scenes):
    """The bookkeeping the kernels' scans rest on (emu_shadow.hip returns an error otherwise): the last list of each kind stays empty,
    no triangle is in both kinds; and the chessboard's squares do go through the coarse bands (far fewer entries than rows / 2)."""
    hs, osc, streams = scenes("chessboard.tri")
    m, om, st = both(oracle, hs, osc, streams, (3.394, 3.394, 4.8), 1024)
    assert st["list_entries"] < 4 * st["drawn"]
    assert st["max_entries_per_tile"] < st["drawn"] // 8


@pytest.mark.parametrize("size,xspan,n", [(8192, 3.99, 300), (1024, 3.99, 2000), (8192, 1.5, 200), (37, 3.9, 6000)])
def test_every_plotted_column_lies_in_the_prep_kernels_range(size, xspan, n):
    """The column pre-filter (sm_prep_projected): triangles given in light space with corners anywhere in (-4 size, 4 size) -- spans
    of up to 8 size serial additions, Light.cc:286-292 -- plot no pixel outside the range c0 .. c1 the prep kernel stores for them,
    and the test is not vacuous: many ranges are narrower than the map, and the spans are as long as the map is wide."""
    tot = dict(drawn=0, narrowed=0, pixels=0, outside=0, longest_span=0)
    for seed in (1, 2):
        r = emu.shadow_column_bound(size, xspan, seed, n)
        for k in tot:
            tot[k] = max(tot[k], r[k]) if k == "longest_span" else tot[k] + r[k]
    assert tot["outside"] == 0, tot
    assert tot["drawn"] > n and tot["narrowed"] > tot["drawn"] // 8 and tot["pixels"] > 100 * n, tot
    assert tot["longest_span"] >= size - 1, tot
