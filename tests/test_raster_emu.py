"""The tiled rasterizer's kernel bodies (renderer_amd/csrc/rs_core.h, ff_add.h) on the CPU.

tests/emu compiles the SAME source the HIP kernels k_rs_setup / k_rs_fill / k_rs_tile are made of for the host and
runs a block as a loop over its threads between the barriers.  These tests compare its frames and counters with the
oracle -- the `-m gpu` suite repeats them on the device through the C ABI; this file is what catches a logic error
before a frame reaches a GPU.  ff_add (the exact fast-forward of the reference's serial `x += d` chains) is checked
against the plain loop."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import renderer_amd as R
from emu import emu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def host_scene():
    cache = {}

    def get(name):
        if name not in cache:
            s = R.Scene(R.assets.mesh_path(name))
            cache[name] = (s, emu.scene_streams(s))
        return cache[name]
    return get


def frames(oracle, oracle_scene, host_scene, mesh, mode, W, H, frame=0, two=False, n_frames=1, **optkw):
    hs, streams = host_scene(mesh)
    osc = oracle_scene(mesh)
    cams, lights, refs, stats = [], [], [], []
    maps = None
    for f in range(n_frames):
        cam, l, n = R.benchmark_frame(frame + f, two)
        ocam, ol, on = oracle.benchmark_frame(frame + f, two)
        if mode in (7, 8) and maps is None:
            maps = [osc.shadowmap(ol[i]) for i in range(on)]          # the benchmark's lights do not move
        ref, _, ost = osc.render(mode, ocam, ol, on, oracle.default_opts(W, H), shadow_maps=maps)
        cams.append(cam); lights.append(l); refs.append(ref); stats.append(ost)
    ho = R.default_opts(W, H, collect_stats=1 if n_frames == 1 else 0, **optkw)
    outs, st, over = emu.render(hs, mode, cams, lights, n, ho, maps, streams=streams)
    assert over == 0
    return outs, refs, st, stats


@pytest.mark.parametrize("mesh,mode,W,H,two", [
    ("chessboard.tri", 6, 1920, 1080, False),          # BASELINE configs[1]
    ("chessboard.tri", 8, 1920, 1080, True),
    ("dragon_vis.ply", 4, 800, 600, False),
    ("dragon_vis.ply", 5, 333, 217, True),
    ("statue.ply", 7, 640, 480, False),
    ("chessboard.tri", 6, 17, 5, False),
    ("chessboard.tri", 8, 1, 1, False),
])
def test_emulated_tiles_equal_the_oracle(oracle, oracle_scene, host_scene, mesh, mode, W, H, two):
    outs, refs, st, ost = frames(oracle, oracle_scene, host_scene, mesh, mode, W, H, frame=3, two=two)
    assert np.array_equal(outs[0], refs[0]), "%d pixels differ" % int((outs[0] != refs[0]).sum())
    assert (st["tris_drawn"], st["spans"], st["ztests"]) == (ost[0].tris_drawn, ost[0].spans, ost[0].ztests)
    assert st["plots"] == int((refs[0] != 0).sum()) or mode in (4, 5)        # winners only (black winners are possible in 4/5)


def test_emulated_batch_equals_single_frames(oracle, oracle_scene, host_scene):
    outs, refs, _, _ = frames(oracle, oracle_scene, host_scene, "chessboard.tri", 8, 640, 360, frame=10, n_frames=5)
    for f in range(5):
        assert np.array_equal(outs[f], refs[f]), "frame %d" % f


def test_emulated_bands(oracle, oracle_scene, host_scene):
    W, H = 640, 360
    hs, streams = host_scene("dragon_vis.ply")
    osc = oracle_scene("dragon_vis.ply")
    cam, l, n = R.benchmark_frame(0)
    ocam, ol, on = oracle.benchmark_frame(0)
    ref, _, _ = osc.render(6, ocam, ol, on, oracle.default_opts(W, H))
    for rows, count in ((8, 2), (15, 4), (16, 3), (5, 8)):
        got = np.zeros_like(ref)
        for index in range(count):
            ho = R.default_opts(W, H, band_rows=rows, band_index=index, band_count=count, compact_rows=1)
            outs, _, over = emu.render(hs, 6, [cam], [l], n, ho, streams=streams)
            sel = [y for y in range(H) if (y // rows) % count == index]
            assert outs[0].shape[0] == len(sel)
            got[sel] = outs[0]
        assert np.array_equal(got, ref), "bands of %d rows over %d ranks" % (rows, count)


def test_small_bins_report_the_overflow(oracle_scene, host_scene):
    hs, streams = host_scene("chessboard.tri")
    cam, l, n = R.benchmark_frame(0)
    outs, _, over = emu.render(hs, 6, [cam], [l], n, R.default_opts(320, 240), bins_cap=1000, streams=streams)
    assert over > 0


def test_ff_add_equals_the_serial_chain(tmp_path):
    """x after k additions of d: ff_add's binade walk against the loop, on random and adversarial operands."""
    src = tmp_path / "ff.cc"
    src.write_text(r'''
#include <math.h>
#include <string.h>
#define MI_HD static inline
#include "%s/renderer_amd/csrc/ff_add.h"
extern "C" void ff_many(const float *x, const float *d, const int *k, float *fast, float *slow, int n)
{
    for (int i = 0; i < n; i++) {
        fast[i] = ff_add(x[i], d[i], k[i]);
        // (ff_add2: two chains of one length; here this chain and its neighbour's operands)
        const int i1 = i + 1 < n ? i + 1 : 0;
        float a = x[i], b = x[i1];
        ff_add2(a, d[i], b, d[i1], k[i]);
        if (memcmp(&a, &fast[i], 4) != 0 && !(a != a && fast[i] != fast[i])) fast[i] = __builtin_nanf("");      // (reported as a difference below)
        const float b1 = ff_add(x[i1], d[i1], k[i]);
        if (memcmp(&b, &b1, 4) != 0 && !(b != b && b1 != b1)) fast[i] = __builtin_nanf("");
        volatile float v = x[i];
        for (int j = 0; j < k[i]; j++) v = v + d[i];
        slow[i] = v;
    }
}
''' % os.path.dirname(HERE))
    so = str(tmp_path / "ff.so")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-o", so, str(src)])
    # ... and with the device's way to the step count (a reciprocal instead of a division), the reciprocal one ulp too large / too small
    libs = [C.CDLL(so)]
    for tag, rcp in (("up", "nextafterf(1.0f / (x), INFINITY)"), ("down", "nextafterf(1.0f / (x), 0.0f)")):
        so2 = str(tmp_path / ("ff_%s.so" % tag))
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-DFF_ADD_TEST_RCP(x)=" + rcp, "-o", so2, str(src)])
        libs.append(C.CDLL(so2))
    rng = np.random.default_rng(2024)
    f = np.float32
    xs, ds, ks = [], [], []
    n = 60000
    # interpolation-like chains (screen coordinates, depths, normals), long ones included
    a = rng.uniform(-3000, 3000, n).astype(f); b = rng.uniform(-3000, 3000, n).astype(f); steps = rng.integers(1, 16384, n)
    xs.append(a); ds.append(((b - a) / steps.astype(f)).astype(f)); ks.append((rng.random(n) * (steps + 1)).astype(np.int32))
    # arbitrary exponents, related by -30..+9 binades
    ex = rng.integers(1, 254, n); off = rng.integers(-30, 10, n)
    mk = lambda e: ((rng.integers(0, 1 << 23, n, dtype=np.uint32) | (np.clip(e, 0, 254).astype(np.uint32) << 23) | (rng.integers(0, 2, n, dtype=np.uint32) << 31))).view(f)
    xs.append(mk(ex)); ds.append(mk(ex + off)); ks.append(rng.integers(0, 4000, n).astype(np.int32))
    # ties: d = (Q + 1/2) ulp(x); binade edges
    ex = rng.integers(30, 220, n); x = mk(ex); u = np.ldexp(f(1), ex - 127 - 23).astype(f)
    xs.append(x); ds.append(((rng.integers(-32, 32, n) + 0.5) * u).astype(f)); ks.append(rng.integers(0, 6000, n).astype(np.int32))
    x = (np.ldexp(f(1), ex - 127) * (1 - rng.integers(0, 3, n) * f(5.9604645e-8))).astype(f)
    xs.append(x); ds.append((np.ldexp(rng.integers(-1000, 1000, n).astype(f), ex - 127 - 23 - rng.integers(0, 4, n))).astype(f)); ks.append(rng.integers(0, 4000, n).astype(np.int32))
    # denormals, zero crossings, overflow, specials
    xs.append(mk(rng.integers(0, 4, n))); ds.append(mk(rng.integers(0, 3, n))); ks.append(rng.integers(0, 4000, n).astype(np.int32))
    x = mk(rng.integers(100, 140, n)); xs.append(x); ds.append((-x / rng.integers(1, 900, n).astype(f) * (1 + rng.random(n).astype(f) * 2)).astype(f)); ks.append(rng.integers(0, 3000, n).astype(np.int32))
    xs.append(mk(rng.integers(250, 255, n))); ds.append(mk(rng.integers(228, 255, n))); ks.append(rng.integers(0, 3000, n).astype(np.int32))
    sp = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 3.4028235e38, -3.4028235e38, 1e-45, -1e-45, 1.17549435e-38, 16777216.0, 8388608.0], f)
    g = np.array(np.meshgrid(sp, sp, [0, 1, 2, 3, 6, 7, 8, 50, 5000])).reshape(3, -1)
    xs.append(g[0].astype(f)); ds.append(g[1].astype(f)); ks.append(g[2].astype(np.int32))
    x, d, k = np.concatenate(xs), np.concatenate(ds), np.concatenate(ks)
    for lib in libs:
        fast, slow = np.empty_like(x), np.empty_like(x)
        lib.ff_many(C.c_void_p(x.ctypes.data), C.c_void_p(d.ctypes.data), C.c_void_p(k.ctypes.data), C.c_void_p(fast.ctypes.data),
                    C.c_void_p(slow.ctypes.data), C.c_int(len(x)))
        same = (fast.view(np.uint32) == slow.view(np.uint32)) | (np.isnan(fast) & np.isnan(slow))
        assert same.all(), "%d of %d chains differ, e.g. x=%r d=%r k=%d" % (int((~same).sum()), len(x), x[~same][0], d[~same][0], k[~same][0])


@pytest.mark.parametrize("nt", [64, 128, 512])
def test_threads_per_tile_do_not_change_the_frame(oracle, oracle_scene, host_scene, nt):
    """mi355_opts::tune[3] = threads of a tile's block, 64..512.  A chunk of the tile's list has RS_CHUNK = 256 slots whatever the
    block's size: round 4 staged one slot per THREAD, so a block of 64 or 128 threads dropped the triangles behind its first 64 /
    128 list entries (found in round 5 by a sweep whose frames differed).  A frame dense enough for tiles with > 64 triangles."""
    outs, refs, st, stats = frames(oracle, oracle_scene, host_scene, "chessboard.tri", 6, 960, 540, tune=R.tune(rsnt=nt))
    assert np.array_equal(outs[0], refs[0])
    assert st["ztests"] == stats[0].ztests
