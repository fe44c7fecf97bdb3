"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package renderer_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")


class Camera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("mv", C.c_float * 9)]


class Light(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("in_camera_space", C.c_float * 3),
                ("camera_to_light", C.c_float * 9), ("world_to_light", C.c_float * 9)]


class Opts(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("screen_dist", C.c_int32),
                ("max_ray_depth", C.c_int32), ("use_shadows", C.c_int32),
                ("use_reflections", C.c_int32), ("antialias", C.c_int32),
                ("shadowmap_size", C.c_int32), ("reflect_rate", C.c_float), ("nudge", C.c_float),
                ("ambient", C.c_float), ("diffuse", C.c_float), ("specular", C.c_float),
                ("clip_z", C.c_float), ("band_rows", C.c_int32), ("band_index", C.c_int32),
                ("band_count", C.c_int32), ("threads", C.c_int32),
                ("use_refractions", C.c_int32), ("refract_rate", C.c_float), ("ambient_occlusion", C.c_int32),
                ("ao_samples", C.c_int32), ("ao_range", C.c_float)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("normal_rays", "shadow_rays", "node_pops", "inner_box_hits", "tri_tests",
                 "plane_pass", "shaded_hits", "max_stack", "tris_drawn", "spans", "ztests", "plots")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build(force: bool = False) -> str:
    """Compile liboracle.so with the pinned strict-IEEE flags (oracle/Makefile)."""
    if force or not os.path.exists(_LIB) or any(
            os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB)
            for f in ("oracle.cc", "oracle.h", "Makefile")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = C.CDLL(_LIB)
        L.orc_scene_load.restype = C.c_void_p
        L.orc_scene_load.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.orc_scene_free.argtypes = [C.c_void_p]
        for f in ("orc_num_vertices", "orc_num_triangles", "orc_bvh_build", "orc_bvh_num_nodes",
                  "orc_bvh_max_depth"):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = C.c_int
        L.orc_bvh_load.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_bvh_save.argtypes = [C.c_void_p, C.c_char_p]
        L.orc_export_vertices.argtypes = [C.c_void_p] + [C.c_void_p] * 3
        L.orc_export_triangles.argtypes = [C.c_void_p] + [C.c_void_p] * 7
        L.orc_bvh_export.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_default_opts.argtypes = [C.POINTER(Opts), C.c_int, C.c_int]
        L.orc_camera_set.argtypes = [C.POINTER(Camera), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_light_update.argtypes = [C.POINTER(Light), C.POINTER(Camera)]
        L.orc_benchmark_frame.argtypes = [C.c_int, C.c_int, C.POINTER(Camera), C.POINTER(Light),
                                          C.POINTER(C.c_int)]
        L.orc_shadowmap_render.argtypes = [C.c_void_p, C.POINTER(Light), C.c_int, C.c_void_p]
        L.orc_render.argtypes = [C.c_void_p, C.c_int, C.POINTER(Camera), C.POINTER(Light), C.c_int,
                                 C.c_void_p, C.POINTER(Opts), C.c_void_p, C.c_int, C.c_void_p,
                                 C.POINTER(Stats)]
        L.orc_render.restype = C.c_int
        L.orc_lighting.argtypes = [C.POINTER(Light), C.c_int, C.c_void_p, C.POINTER(Opts), C.c_int, C.c_int,
                                   C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def default_opts(width: int, height: int, **kw) -> Opts:
    o = Opts()
    lib().orc_default_opts(C.byref(o), width, height)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def wu_lines(pixels: np.ndarray, xyxy) -> np.ndarray:
    """orc_wu_lines: draw the lines (n x 4 int16: x1, y1, x2, y2) over `pixels` (H x W uint32, modified in place)"""
    xyxy = np.ascontiguousarray(xyxy, np.int16).reshape(-1, 4)
    f = lib().orc_wu_lines
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    f.restype = None
    f(pixels.ctypes.data, pixels.shape[1], pixels.shape[0], pixels.strides[0] // 4, len(xyxy), xyxy.ctypes.data)
    return pixels


def benchmark_frame(k: int, second_light: bool = False):
    """Camera + lights of frame k of the reference's `renderer -b` orbit."""
    cam = Camera()
    lights = (Light * 2)()
    n = C.c_int(0)
    lib().orc_benchmark_frame(k, int(second_light), C.byref(cam), lights, C.byref(n))
    return cam, lights, n.value


def camera(eye, lookat) -> Camera:
    """Camera at `eye` looking at `lookat` (Camera.cc:24-42)."""
    cam = Camera()
    lib().orc_camera_set(C.byref(cam), (C.c_float * 3)(*eye), (C.c_float * 3)(*lookat))
    return cam


def light(pos, cam: Camera) -> Light:
    """Light at `pos` with its camera-space members derived for `cam` (Light.cc:162-216)."""
    l = Light()
    l.pos[:] = list(pos)
    lib().orc_light_update(C.byref(l), C.byref(cam))
    return l


class Scene:
    def __init__(self, path: str):
        err = C.create_string_buffer(256)
        self._h = lib().orc_scene_load(path.encode(), err, 256)
        if not self._h:
            raise RuntimeError(err.value.decode())
        self.path = path
        self.nv = lib().orc_num_vertices(self._h)
        self.nt = lib().orc_num_triangles(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_scene_free(self._h)
            self._h = None

    # ---- data exports -----------------------------------------------------
    def vertices(self):
        vpos = np.empty((self.nv, 3), np.float32)
        vnrm = np.empty((self.nv, 3), np.float32)
        vao = np.empty(self.nv, np.uint32)
        lib().orc_export_vertices(self._h, vpos.ctypes.data, vnrm.ctypes.data, vao.ctypes.data)
        return vpos, vnrm, vao

    def triangles(self):
        T = self.nt
        out = dict(idx=np.empty((T, 3), np.int32), center=np.empty((T, 3), np.float32),
                   normal=np.empty((T, 3), np.float32), colorf=np.empty((T, 3), np.float32),
                   color32=np.empty(T, np.uint32), two_sided=np.empty(T, np.uint8),
                   plane=np.empty((T, 16), np.float32))
        lib().orc_export_triangles(self._h, *[out[k].ctypes.data for k in
                                              ("idx", "center", "normal", "colorf", "color32",
                                               "two_sided", "plane")])
        return out

    # ---- BVH ---------------------------------------------------------------
    def bvh_build(self) -> int:
        n = lib().orc_bvh_build(self._h)
        if n < 0:
            raise RuntimeError("oracle BVH build failed (%d)" % n)
        return n

    def bvh_load(self, path: str) -> int:
        return lib().orc_bvh_load(self._h, path.encode())

    def bvh_save(self, path: str) -> int:
        return lib().orc_bvh_save(self._h, path.encode())

    def bvh_ensure(self, cache_path: str | None = None) -> int:
        """Load the oracle's own .bvh cache if present, else build and save it."""
        if cache_path and os.path.exists(cache_path) and self.bvh_load(cache_path) > 0:
            return self.num_nodes
        n = self.bvh_build()
        if cache_path:
            self.bvh_save(cache_path)
        return n

    @property
    def num_nodes(self) -> int:
        return lib().orc_bvh_num_nodes(self._h)

    @property
    def max_depth(self) -> int:
        return lib().orc_bvh_max_depth(self._h)

    def bvh(self):
        n = self.num_nodes
        nodes = np.empty((n, 8), np.uint32)
        tri_idx = np.empty(self.nt, np.int32)
        lib().orc_bvh_export(self._h, nodes.ctypes.data, tri_idx.ctypes.data)
        return nodes, tri_idx

    # ---- rendering ---------------------------------------------------------
    def shadowmap(self, light: Light, size: int = 1024) -> np.ndarray:
        m = np.empty((size, size), np.float32)
        lib().orc_shadowmap_render(self._h, C.byref(light), size, m.ctypes.data)
        return m

    def trace_hits(self, rays):
        """orc_trace_hits: rays (n, 6) float32 -> (triangle index or -1 [n], hit point [n, 3])"""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        tri = np.zeros(len(rays), np.int32)
        hit = np.zeros((len(rays), 3), np.float32)
        f = lib().orc_trace_hits
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        f.restype = None
        f(self._h, len(rays), rays.ctypes.data, tri.ctypes.data, hit.ctypes.data)
        return tri, hit

    def render(self, mode: int, cam: Camera, lights, n_lights: int, opts: Opts, shadow_maps=None,
               want_f32: bool = False):
        W, H = opts.width, opts.height
        out = np.zeros((H, W), np.uint32)
        outf = np.zeros((H, W, 3), np.float32) if want_f32 else None
        st = Stats()
        maps_arg = None
        if shadow_maps is not None:
            arr = (C.c_void_p * len(shadow_maps))(*[m.ctypes.data for m in shadow_maps])
            maps_arg = C.cast(arr, C.c_void_p)
        rc = lib().orc_render(self._h, mode, C.byref(cam), lights, n_lights, maps_arg, C.byref(opts),
                              out.ctypes.data, W, outf.ctypes.data if want_f32 else None, C.byref(st))
        if rc != 0:
            raise RuntimeError("orc_render failed (%d)" % rc)
        return out, outf, st


def raster_winners(scene, mode: int, cam: Camera, lights, n_lights: int, opts: Opts, shadow_maps=None):
    """orc_raster_winners: (frame, winning triangle per pixel or -1, Z-passes per pixel, the fat point Plot<> received [H, W, 8])."""
    W, H = opts.width, opts.height
    out = np.zeros((H, W), np.uint32)
    tri = np.zeros((H, W), np.int32)
    passes = np.zeros((H, W), np.int32)
    fat = np.zeros((H, W, 8), np.float32)
    maps_arg = None
    if shadow_maps is not None:
        arr = (C.c_void_p * len(shadow_maps))(*[m.ctypes.data for m in shadow_maps])
        maps_arg = C.cast(arr, C.c_void_p)
    f = lib().orc_raster_winners
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    f.restype = C.c_int
    rc = f(scene._h, mode, C.byref(cam), lights, n_lights, maps_arg, C.byref(opts), out.ctypes.data, tri.ctypes.data, passes.ctypes.data, fat.ctypes.data)
    if rc != 0:
        raise RuntimeError("orc_raster_winners failed (%d)" % rc)
    return out, tri, passes, fat


def lighting(lights, n_lights: int, opts: Opts, shadow_mode: int, points, shadow_maps=None) -> np.ndarray:
    """LightingEquation<mode>::ComputePixel for rows (inCameraSpace[3], normal[3], material r,g,b, ao)."""
    pts = np.ascontiguousarray(points, np.float32)
    out = np.empty((len(pts), 3), np.float32)
    maps_arg = None
    if shadow_maps is not None:
        arr = (C.c_void_p * len(shadow_maps))(*[m.ctypes.data for m in shadow_maps])
        maps_arg = C.cast(arr, C.c_void_p)
    lib().orc_lighting(lights, n_lights, maps_arg, C.byref(opts), shadow_mode, len(pts), pts.ctypes.data, out.ctypes.data)
    return out


def mlaa(pixels: np.ndarray) -> np.ndarray:
    """MLAA(fbi, NULL, width, height) of the reference on a copy of `pixels` (uint32 [H, W])."""
    out = np.ascontiguousarray(pixels, np.uint32).copy()
    if lib().orc_mlaa(C.c_void_p(out.ctypes.data), C.c_int(out.shape[1]), C.c_int(out.shape[0])) != 0:
        raise RuntimeError("orc_mlaa: frame size %dx%d is not one the reference's MLAA handles" % (out.shape[1], out.shape[0]))
    return out


def rgb_bytes(xrgb: np.ndarray) -> bytes:
    """Raw R,G,B bytes (row-major, top row first) -- the layout SURVEY.md 8(c) hashes."""
    a = np.ascontiguousarray(xrgb)
    rgb = np.stack([(a >> 16) & 0xff, (a >> 8) & 0xff, a & 0xff], axis=-1).astype(np.uint8)
    return rgb.tobytes()
