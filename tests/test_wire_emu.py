"""The wireframe's line generator (renderer_amd/csrc/wf_core.h, what k_wire.hip expands lines with) compiled for the HOST,
against the oracle's separate restatement of my_aalineColor and the blending pixel routines of Wu.cc: same pixels for lines
of every slope, length and position -- inside, across and far outside the surface, coordinates that wrapped to negative
16-bit values, degenerate lines -- drawn over each other in order.  (Both are restatements of Wu.cc as read: see the note
on unpinned parity in oracle/oracle.cc.)"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "emu"), "-s", "libemu_wire.so"])
    L = C.CDLL(os.path.join(HERE, "emu", "libemu_wire.so"))
    L.emu_wire_lines.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.emu_wire_lines.restype = None

    L.emu_wire_lines_table.argtypes = L.emu_wire_lines.argtypes
    L.emu_wire_lines_table.restype = None
    L.emu_wire_check_table.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.emu_wire_check_table.restype = C.c_int

    def draw(pixels, xyxy, table=False):
        xyxy = np.ascontiguousarray(xyxy, np.int16).reshape(-1, 4)
        (L.emu_wire_lines_table if table else L.emu_wire_lines)(pixels.ctypes.data, pixels.shape[1], pixels.shape[0], pixels.strides[0] // 4, len(xyxy), xyxy.ctypes.data)
        return pixels

    def check_table(W, H, xyxy):
        xyxy = np.ascontiguousarray(xyxy, np.int16).reshape(-1, 4)
        return L.emu_wire_check_table(W, H, len(xyxy), xyxy.ctypes.data)
    draw.check_table = check_table
    return draw


def lines(rng, n, W, H, kind):
    if kind == "inside":
        return np.stack([rng.integers(0, W, n), rng.integers(0, H, n), rng.integers(0, W, n), rng.integers(0, H, n)], 1)
    if kind == "around":
        return np.stack([rng.integers(-W, 2 * W, n), rng.integers(-H, 2 * H, n), rng.integers(-W, 2 * W, n), rng.integers(-H, 2 * H, n)], 1)
    if kind == "wild":
        return rng.integers(-32768, 32768, (n, 4))
    if kind == "axis":      # horizontal, vertical, diagonal, single pixels
        a = np.stack([rng.integers(-20, W + 20, n), rng.integers(-20, H + 20, n)], 1)
        d = rng.integers(-60, 61, n)
        pick = rng.integers(0, 5, n)
        b = a.copy()
        b[pick == 0, 0] += d[pick == 0]
        b[pick == 1, 1] += d[pick == 1]
        b[pick == 2] += np.stack([d, d], 1)[pick == 2]
        b[pick == 3] += np.stack([d, -d], 1)[pick == 3]
        return np.concatenate([a, b], 1)
    raise ValueError(kind)


@pytest.mark.parametrize("kind", ["inside", "around", "wild", "axis"])
@pytest.mark.parametrize("W,H", [(64, 48), (333, 187), (1920, 1080), (17, 9), (1, 1)])
def test_generator_draws_what_the_oracle_draws(oracle, emu, kind, W, H):
    rng = np.random.default_rng(hash((kind, W, H)) & 0xffff)
    xyxy = lines(rng, 4000 if W < 500 else 1500, W, H, kind).astype(np.int16)
    a = oracle.wu_lines(np.zeros((H, W), np.uint32), xyxy)
    b = emu(np.zeros((H, W), np.uint32), xyxy)
    assert (a != 0).sum() > 0 or kind == "wild"
    bad = np.argwhere(a != b)
    assert bad.size == 0, "%d pixels differ, first at %s: %#x vs %#x" % (len(bad), bad[0], a[tuple(bad[0])], b[tuple(bad[0])])


def test_one_line_at_a_time(oracle, emu):
    """(so that a mismatch names the line)"""
    rng = np.random.default_rng(5)
    W, H = 97, 61
    for kind in ("inside", "around", "axis", "wild"):
        for l in lines(rng, 1500, W, H, kind).astype(np.int16):
            a = oracle.wu_lines(np.zeros((H, W), np.uint32), l)
            b = emu(np.zeros((H, W), np.uint32), l)
            assert np.array_equal(a, b), "line %s (%s)" % (l.tolist(), kind)


@pytest.mark.parametrize("kind", ["inside", "around", "wild", "axis"])
@pytest.mark.parametrize("W,H", [(64, 48), (333, 187), (1920, 1080), (4095, 2048), (17, 9), (1, 1)])
def test_the_table_form_is_the_walk_call_by_call(oracle, emu, kind, W, H):
    """k_wire.hip expands a line with one thread per emit call (wf_plan / wf_op): every call of the walk, the ones that fall
    outside the surface included, is the table's entry of the same index -- and the frame drawn through the table is the
    oracle's"""
    rng = np.random.default_rng(hash(("table", kind, W, H)) & 0xffff)
    xyxy = lines(rng, 6000 if W < 500 else 2500, W, H, kind).astype(np.int16)
    bad = emu.check_table(W, H, xyxy)
    assert bad < 0, "line %d: %s" % (bad, xyxy[bad].tolist())
    if W * H <= 1920 * 1080:
        a = oracle.wu_lines(np.zeros((H, W), np.uint32), xyxy)
        b = emu(np.zeros((H, W), np.uint32), xyxy, table=True)
        assert np.array_equal(a, b)


def test_the_table_on_every_short_line(emu):
    """all lines between points of a small grid that reaches past the surface on every side: every slope, both directions,
    every clip case"""
    W, H = 13, 11
    pts = [(x, y) for x in range(-4, W + 4) for y in range(-4, H + 4)]
    xyxy = np.array([(a[0], a[1], b[0], b[1]) for a in pts for b in pts], np.int16)
    bad = emu.check_table(W, H, xyxy)
    assert bad < 0, "line %s" % xyxy[bad].tolist()
