"""ctypes side of tests/emu/libemu_raster.so: the tiled rasterizer's kernel bodies compiled for the host (TEST INFRASTRUCTURE)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

import renderer_amd as R

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libemu_raster.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
        _lib = C.CDLL(_LIB)
        _lib.emu_raster.restype = C.c_int
    return _lib


def scene_streams(hs):
    """DevScene's rs_* streams, packed as mi355_scene_create packs them (renderer_amd/csrc/capi.hip)."""
    a = hs.arrays()
    T, V = hs.nt, hs.nv
    rs_tri = np.zeros((T, 2, 4), np.float32)
    rs_tri[:, 0, :3] = a["tri_center"]
    rs_tri[:, 0, 3] = (a["tri_two_sided"] != 0).astype(np.uint32).view(np.float32)
    rs_tri[:, 1, :3] = a["tri_normal"]
    rs_tri[:, 1, 3] = a["tri_color32"].view(np.float32)
    rs_col = np.zeros((T, 4), np.float32)
    rs_col[:, :3] = a["tri_colorf"]
    rs_idx = np.zeros((T, 4), np.uint32)
    rs_idx[:, :3] = a["tri_index"]
    rs_vert = np.zeros((V, 2, 4), np.float32)
    rs_vert[:, 0, :3] = a["vertex_pos"]
    rs_vert[:, 0, 3] = a["vertex_ao"].astype(np.float32)
    rs_vert[:, 1, :3] = a["vertex_normal"]
    return rs_tri, rs_col, rs_idx, rs_vert


def render(hs, mode, cams, lights_per_frame, n_lights, opts, shadow_maps=None, bins_cap=0, streams=None, band_cap=0):
    """Frames of `cams` (a list) -> (list of images, stats dict, dropped bin entries)."""
    rs_tri, rs_col, rs_idx, rs_vert = streams or scene_streams(hs)
    n = len(cams)
    W, H = opts.width, opts.height
    rows = H
    if opts.band_count > 1 and opts.compact_rows:
        rows = sum(1 for y in range(H) if (y // opts.band_rows) % opts.band_count == opts.band_index)
    outs = [np.full((rows, W), 0xabababab, np.uint32) for _ in range(n)]
    cam_arr = (R.Camera * n)(*cams)
    light_arr = (R.Light * max(1, n * n_lights))()
    for f in range(n):
        for i in range(n_lights):
            light_arr[f * n_lights + i] = lights_per_frame[f][i]
    maps = None
    if shadow_maps is not None:
        maps = (C.c_void_p * len(shadow_maps))(*[m.ctypes.data for m in shadow_maps])
    out_ptrs = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    stats = (C.c_ulonglong * 4)()
    over = C.c_ulonglong(0)
    rc = lib().emu_raster(C.c_uint32(hs.nt), C.c_uint32(hs.nv), C.c_void_p(rs_tri.ctypes.data), C.c_void_p(rs_col.ctypes.data),
                          C.c_void_p(rs_idx.ctypes.data), C.c_void_p(rs_vert.ctypes.data), C.c_int(mode), C.c_int(n), cam_arr, light_arr,
                          C.c_int(n_lights), C.byref(opts), maps, out_ptrs, C.c_int(W), stats, C.c_uint32(bins_cap), C.byref(over), C.c_uint32(band_cap))
    if rc != 0:
        raise RuntimeError("emu_raster failed (%d)" % rc)
    return outs, dict(tris_drawn=stats[0], spans=stats[1], ztests=stats[2], plots=stats[3]), int(over.value)


_slib = None


def shadowmap(hs, light, size=1024, streams=None, items_per_tile=None):
    """tests/emu/libemu_shadow.so: the shadow map through the kernels' own per-thread code (sm_core.h) on the host -> (map, stats)."""
    global _slib
    if _slib is None:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
        _slib = C.CDLL(os.path.join(_HERE, "libemu_shadow.so"))
        _slib.emu_shadowmap.restype = C.c_int
    _, _, rs_idx, rs_vert = streams or scene_streams(hs)
    out = np.empty((size, size), np.float32)
    st = (C.c_ulonglong * 6)()
    pos = (C.c_float * 3)(*list(light.pos))
    w2l = (C.c_float * 9)(*list(light.world_to_light))
    rc = _slib.emu_shadowmap(C.c_uint32(hs.nt), C.c_uint32(hs.nv), C.c_void_p(rs_idx.ctypes.data), C.c_void_p(rs_vert.ctypes.data), pos, w2l,
                             C.c_int(size), C.c_void_p(out.ctypes.data), st, C.c_void_p(items_per_tile.ctypes.data if items_per_tile is not None else None))
    if rc != 0:
        raise RuntimeError("emu_shadowmap failed (%d)" % rc)
    return out, dict(drawn=st[0], list_entries=st[1], tile_entries=st[2], items=st[3], max_entries_per_tile=st[4], pixels=st[5])


def shadow_column_bound(size, xspan, seed, n):
    """tests/emu/libemu_shadow.so: random light-space triangles against the prep kernel's column range (emu_shadow_column_bound)."""
    global _slib
    if _slib is None:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
        _slib = C.CDLL(os.path.join(_HERE, "libemu_shadow.so"))
        _slib.emu_shadowmap.restype = C.c_int
    out = (C.c_ulonglong * 5)()
    _slib.emu_shadow_column_bound.restype = C.c_int
    rc = _slib.emu_shadow_column_bound(C.c_int(size), C.c_float(xspan), C.c_uint32(seed), C.c_int(n), out)
    if rc != 0:
        raise RuntimeError("emu_shadow_column_bound failed (%d)" % rc)
    return dict(drawn=out[0], narrowed=out[1], pixels=out[2], outside=out[3], longest_span=out[4])
