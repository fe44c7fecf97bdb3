"""renderer_amd -- MI355X-native hot path of ttsiodras/renderer.

The product is native: HIP kernels + a C ABI (``include/mi355_render.h``,
``lib/libmi355render.so``) and a C++ host layer with the reference's
Scene/Camera/Light/Screen API (``csrc/host``, ``lib/libmi355host.so``).  This
Python module is only the ctypes glue the tests, ``bench.py`` and the
multi-GPU driver (``renderer_amd.multigpu``) use to call them; it contains no
rendering code and no CPU fallback -- if the native libraries are missing or
no GPU is present, rendering calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import subprocess

import numpy as np

from . import assets  # noqa: F401

_PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_PKG)
LIB_DIR = os.path.join(_PKG, "lib")
# (MI355_RENDER_SO: another build of the same library -- measurement scripts compare kernel variants that way)
RENDER_SO = os.environ.get("MI355_RENDER_SO") or os.path.join(LIB_DIR, "libmi355render.so")
HOST_SO = os.path.join(LIB_DIR, "libmi355host.so")
RENDER_CLI = os.path.join(LIB_DIR, "render_cli")
HEADER = os.path.join(ROOT, "include", "mi355_render.h")

MAX_LIGHTS = 4


MAX_IN_FLIGHT = 4      # MI355_MAX_IN_FLIGHT (include/mi355_render.h)


class Mi355Error(RuntimeError):
    pass


# ---------------------------------------------------------------- ABI structs (include/mi355_render.h)
class Camera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("mv", C.c_float * 9)]


class Light(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("in_camera_space", C.c_float * 3),
                ("camera_to_light", C.c_float * 9), ("world_to_light", C.c_float * 9)]


class Opts(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("screen_dist", C.c_int32),
                ("max_ray_depth", C.c_int32), ("use_shadows", C.c_int32), ("use_reflections", C.c_int32),
                ("shadowmap_size", C.c_int32), ("reflect_rate", C.c_float), ("nudge", C.c_float),
                ("ambient", C.c_float), ("diffuse", C.c_float), ("specular", C.c_float),
                ("clip_z", C.c_float), ("band_rows", C.c_int32), ("band_index", C.c_int32),
                ("band_count", C.c_int32), ("compact_rows", C.c_int32), ("collect_stats", C.c_int32),
                ("tune", C.c_int32 * 8), ("mlaa", C.c_int32),
                ("use_refractions", C.c_int32), ("refract_rate", C.c_float), ("ambient_occlusion", C.c_int32),
                ("ao_samples", C.c_int32), ("ao_range", C.c_float), ("keep_canvas", C.c_int32), ("reserved", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("normal_rays", "shadow_rays", "node_pops", "inner_box_hits", "tri_tests", "plane_pass",
                 "shaded_hits", "tris_drawn", "spans", "ztests", "plots")] + \
               [("kernel_ms", C.c_float), ("reserved", C.c_float)]

    def as_dict(self):
        return {n: (int(getattr(self, n)) if t is C.c_uint64 else float(getattr(self, n)))
                for n, t in self._fields_}


class SceneDesc(C.Structure):
    _fields_ = [("n_vertices", C.c_uint32), ("n_triangles", C.c_uint32),
                ("vertex_pos", C.POINTER(C.c_float)), ("vertex_normal", C.POINTER(C.c_float)),
                ("vertex_ao", C.POINTER(C.c_uint32)), ("tri_index", C.POINTER(C.c_int32)),
                ("tri_center", C.POINTER(C.c_float)), ("tri_normal", C.POINTER(C.c_float)),
                ("tri_colorf", C.POINTER(C.c_float)), ("tri_color32", C.POINTER(C.c_uint32)),
                ("tri_two_sided", C.POINTER(C.c_uint8)), ("tri_d", C.POINTER(C.c_float)),
                ("tri_e", C.POINTER(C.c_float))]


# ---------------------------------------------------------------- native libraries
def build(force: bool = False) -> None:
    """Compile the HIP C-ABI library for gfx950 and the C++ host layer (in-tree, under lib/)."""
    args = ["-B"] if force else []
    subprocess.check_call(["make", "-s", "-C", os.path.join(_PKG, "csrc"), "-j4"] + args)
    subprocess.check_call(["make", "-s", "-C", os.path.join(_PKG, "csrc", "host")] + args)


_render = None
_host = None


def lib():
    """libmi355render.so -- the C ABI.  Raises if it has not been built: there is no fallback."""
    global _render
    if _render is None:
        if not os.path.exists(RENDER_SO):
            raise Mi355Error("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(the HIP extension is required, there is no CPU path)" % RENDER_SO)
        # A PyTorch-ROCm wheel carries its own copies of the HIP runtime and RCCL.  If this library (linked against
        # /opt/rocm) is loaded first and torch later, the process holds two runtimes and aborts at exit ("double free",
        # seen on the MI355X boxes); with torch first the loader hands this library the copies already in the process.
        # torch is what the callers here use for device buffers and streams anyway, so it goes first when it exists.
        if "torch" not in sys.modules:
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        L = C.CDLL(RENDER_SO, mode=C.RTLD_GLOBAL)
        L.mi355_abi_version.restype = C.c_int
        L.mi355_init.argtypes = [C.c_int, C.POINTER(C.c_int)]
        L.mi355_last_error.restype = C.c_char_p
        L.mi355_default_opts.argtypes = [C.POINTER(Opts), C.c_int, C.c_int]
        L.mi355_scene_create.restype = C.c_void_p
        L.mi355_scene_create.argtypes = [C.POINTER(SceneDesc), C.c_int]
        L.mi355_scene_destroy.argtypes = [C.c_void_p]
        L.mi355_scene_set_bvh.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.mi355_shadowmap_render.argtypes = [C.c_void_p, C.c_int, C.POINTER(Light), C.c_int, C.c_void_p]
        L.mi355_shadowmap_set.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.mi355_render.argtypes = [C.c_void_p, C.c_int, C.POINTER(Camera), C.POINTER(Light), C.c_int,
                                   C.POINTER(Opts), C.c_void_p, C.c_int, C.c_void_p, C.POINTER(Stats)]
        L.mi355_render_device.argtypes = [C.c_void_p, C.c_int, C.POINTER(Camera), C.POINTER(Light), C.c_int,
                                          C.POINTER(Opts), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.mi355_fetch_stats.argtypes = [C.c_void_p, C.POINTER(Stats)]
        _render = L
    return _render


def host():
    """libmi355host.so -- the C++ Scene/Camera/Light layer (loaders, BVH builder, harness)."""
    global _host
    if _host is None:
        lib()
        if not os.path.exists(HOST_SO):
            raise Mi355Error("%s is missing: run __graft_entry__.build()" % HOST_SO)
        H = C.CDLL(HOST_SO)
        H.mi355h_last_error.restype = C.c_char_p
        H.mi355h_scene_load.restype = C.c_void_p
        H.mi355h_scene_load.argtypes = [C.c_char_p]
        H.mi355h_scene_free.argtypes = [C.c_void_p]
        H.mi355h_dump_3ds.argtypes = [C.c_char_p, C.c_char_p]
        H.mi355h_scene_desc.argtypes = [C.c_void_p, C.POINTER(SceneDesc)]
        H.mi355h_set_device.argtypes = [C.c_void_p, C.c_int]
        H.mi355h_bvh_create.argtypes = [C.c_void_p]
        H.mi355h_bvh_update.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        H.mi355h_bvh_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_int),
                                      C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        H.mi355h_opts.restype = C.POINTER(Opts)
        H.mi355h_opts.argtypes = [C.c_void_p]
        H.mi355h_context.restype = C.c_void_p
        H.mi355h_context.argtypes = [C.c_void_p]
        H.mi355h_camera_set.argtypes = [C.POINTER(Camera), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        H.mi355h_light_update.argtypes = [C.POINTER(Light), C.POINTER(Camera)]
        H.mi355h_benchmark_frame.argtypes = [C.c_int, C.c_int, C.POINTER(Camera), C.POINTER(Light), C.POINTER(C.c_int)]
        H.mi355h_render_frame.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float),
                                          C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.c_void_p,
                                          C.POINTER(Stats)]
        _host = H
    return _host


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise Mi355Error("%s failed (%d): %s" % (what, rc, lib().mi355_last_error().decode()))


def default_opts(width: int, height: int, **kw) -> Opts:
    o = Opts()
    lib().mi355_default_opts(C.byref(o), width, height)
    for k, v in kw.items():
        if k == "tune":
            o.tune[:] = tune(**v) if isinstance(v, dict) else list(v)
        else:
            setattr(o, k, v)
    return o


_EXITING = False


def _mark_exit():
    global _EXITING
    _EXITING = True


import atexit as _atexit  # noqa: E402
_atexit.register(_mark_exit)


def tune(xmin=0, rmin=0, chunk=0, lmin=0, bpc=0, exact=0, rowmajor=0, scatter=0, reforder=0, profordered=0, nohelp=0, nocull=0, nopipe=0, noshare=0, sharemin=0, rsnt=0):
    """mi355_opts::tune as a list (see include/mi355_render.h); every knob leaves the pixels unchanged.
    (lmin, scatter and nohelp are accepted for old scripts and ignored.)"""
    flags = (1 if exact else 0) | (2 if rowmajor else 0) | (4 if reforder else 0) | (8 if profordered else 0) | (16 if nocull else 0) | (32 if nopipe else 0) | (256 if noshare else 0)
    return [xmin, rmin, chunk, rsnt, bpc, flags, sharemin, 0]


def device_count() -> int:
    n = C.c_int(0)
    return n.value if lib().mi355_init(0, C.byref(n)) == 0 else 0


def host_array(shape, dtype=np.uint32) -> np.ndarray:
    """An array in frame memory of the library's (mi355_host_alloc: page-locked from the start, zero-filled): mi355_render writes
    raytraced frames into it directly, mi355_render_async copies with one DMA transfer -- no host_register.  Give it back with
    host_array_free(a) once no view of it is in use."""
    f = lib().mi355_host_alloc
    f.restype = C.c_void_p
    f.argtypes = [C.c_size_t]
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = f(n)
    if not p:
        raise Mi355Error("mi355_host_alloc: " + lib().mi355_last_error().decode())
    a = np.ctypeslib.as_array((C.c_ubyte * n).from_address(p)).view(dtype).reshape(shape)
    _HOST_ARRAYS[p] = n
    return a


_HOST_ARRAYS = {}


def host_array_free(a: np.ndarray) -> None:
    """Give back the allocation `a` lies in -- `a` may be the array host_array() returned or any view of it (a slice, a reshape,
    another dtype): the allocation is found by address.  ValueError for an array that is not in frame memory of the library's."""
    addr = a.ctypes.data
    p = next((b for b, n in _HOST_ARRAYS.items() if b <= addr < b + max(n, 1)), None)
    if p is None:
        raise ValueError("host_array_free: the array at 0x%x does not lie in memory that host_array() handed out (or it was freed already)" % addr)
    del _HOST_ARRAYS[p]
    f = lib().mi355_host_free
    f.restype = None
    f.argtypes = [C.c_void_p]
    f(p)


def own_mapping_array(shape, dtype=np.uint32) -> np.ndarray:
    """An array over an anonymous private mapping of its own (page-aligned, not a piece of the malloc heap): what to hand to
    Scene.host_register when the memory has to be the caller's.  Unmapped when the last reference goes."""
    import mmap
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    mm = mmap.mmap(-1, (n + mmap.PAGESIZE - 1) // mmap.PAGESIZE * mmap.PAGESIZE)
    return np.frombuffer(mm, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def benchmark_frame(k: int, second_light: bool = False):
    """Camera and lights of frame k of the reference's `renderer -b` loop (host C++ harness)."""
    cam = Camera()
    lights = (Light * 2)()
    n = C.c_int(0)
    host().mi355h_benchmark_frame(k, int(second_light), C.byref(cam), lights, C.byref(n))
    return cam, lights, n.value


def camera(eye, lookat) -> Camera:
    cam = Camera()
    host().mi355h_camera_set(C.byref(cam), (C.c_float * 3)(*eye), (C.c_float * 3)(*lookat))
    return cam


def light(pos, cam: Camera) -> Light:
    l = Light()
    l.pos[:] = list(pos)
    host().mi355h_light_update(C.byref(l), C.byref(cam))
    return l


class _OwnedArray(np.ndarray):
    """A view into memory owned by a C++ object: keeps that object alive as long as the view (or a view of it) lives."""
    _owner = None


def _np_view(ptr, n, dtype, owner=None):
    if n == 0:
        return np.zeros(0, dtype)
    a = np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype).view(_OwnedArray)
    a._owner = owner
    return a


def dump_3ds(path_3ds: str, path_out: str) -> None:
    """Test hook: write what the .3ds reader hands to Scene::load (before the loader's common tail) as an .r3ds dump."""
    if host().mi355h_dump_3ds(path_3ds.encode(), path_out.encode()) != 0:
        raise Mi355Error(host().mi355h_last_error().decode())


class Scene:
    """A loaded mesh (C++ mi355::Scene) and, lazily, its device context."""

    def __init__(self, path: str, device: int = 0):
        self._h = host().mi355h_scene_load(path.encode())
        if not self._h:
            raise Mi355Error(host().mi355h_last_error().decode())
        self.path = path
        if device:
            host().mi355h_set_device(self._h, device)
        d = SceneDesc()
        host().mi355h_scene_desc(self._h, C.byref(d))
        self.desc = d
        self.nv, self.nt = int(d.n_vertices), int(d.n_triangles)

    def set_devices(self, devices) -> None:
        """mi355::Scene::_devices: every frame of the C++ render* API on these GPUs (a device listed twice plays two ranks)."""
        arr = (C.c_int * len(devices))(*devices)
        f = host().mi355h_set_devices
        f.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
        f(self._h, arr, len(devices))

    def close(self):
        if getattr(self, "_h", None):
            host().mi355h_scene_free(self._h)
            self._h = None

    def __del__(self):
        # at interpreter exit the HIP runtime may already be gone: freeing device memory then can crash the process
        # after all the work is done, so scenes still alive at exit are left to the OS
        if _EXITING:
            return
        try:
            self.close()
        except Exception:
            pass

    # -- host data (views into the C++ Scene's arrays) ------------------------------------------
    def arrays(self):
        d, V, T = self.desc, self.nv, self.nt
        return dict(
            vertex_pos=_np_view(d.vertex_pos, 3 * V, np.float32, self).reshape(V, 3),
            vertex_normal=_np_view(d.vertex_normal, 3 * V, np.float32, self).reshape(V, 3),
            vertex_ao=_np_view(d.vertex_ao, V, np.uint32, self),
            tri_index=_np_view(d.tri_index, 3 * T, np.int32, self).reshape(T, 3),
            tri_center=_np_view(d.tri_center, 3 * T, np.float32, self).reshape(T, 3),
            tri_normal=_np_view(d.tri_normal, 3 * T, np.float32, self).reshape(T, 3),
            tri_colorf=_np_view(d.tri_colorf, 3 * T, np.float32, self).reshape(T, 3),
            tri_color32=_np_view(d.tri_color32, T, np.uint32, self),
            tri_two_sided=_np_view(d.tri_two_sided, T, np.uint8, self),
            tri_d=_np_view(d.tri_d, 4 * T, np.float32, self).reshape(T, 4),
            tri_e=_np_view(d.tri_e, 9 * T, np.float32, self).reshape(T, 9))

    # -- BVH ---------------------------------------------------------------------------------------
    def bvh_create(self, where: str = "auto") -> int:
        """Scene::CreateBVH.  where: "auto" (GPU builder if a device is usable, else the host builder), "host", "device".
        self.bvh_built_on_device tells which one ran."""
        on = C.c_int(0)
        f = host().mi355h_bvh_create_on
        f.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        if f(self._h, {"auto": 0, "host": 1, "device": 2}[where], C.byref(on)) != 0:
            raise Mi355Error(host().mi355h_last_error().decode())
        self.bvh_built_on_device = bool(on.value)
        return self.bvh_info()[0]

    def bvh_update(self, filename: str | None = None, force: bool = False) -> int:
        """Scene::UpdateBoundingVolumeHierarchy: `<filename>.bvh` cache or build."""
        if host().mi355h_bvh_update(self._h, (filename or self.path).encode(), int(force)) != 0:
            raise Mi355Error(host().mi355h_last_error().decode())
        return self.bvh_info()[0]

    def bvh_info(self):
        nn, ni, md = C.c_uint32(0), C.c_uint32(0), C.c_int(0)
        pn, pi = C.c_void_p(0), C.c_void_p(0)
        host().mi355h_bvh_info(self._h, C.byref(nn), C.byref(ni), C.byref(md), C.byref(pn), C.byref(pi))
        return nn.value, ni.value, md.value, pn.value, pi.value

    def bvh_arrays(self):
        nn, ni, _, pn, pi = self.bvh_info()
        if nn == 0:
            return np.zeros((0, 8), np.uint32), np.zeros(0, np.int32)
        # copies: the C++ vectors behind these pointers are replaced by the next build
        nodes = np.ctypeslib.as_array(C.cast(pn, C.POINTER(C.c_uint32)), shape=(nn * 8,)).reshape(nn, 8).copy()
        idx = np.ctypeslib.as_array(C.cast(pi, C.POINTER(C.c_int32)), shape=(ni,)).copy()
        return nodes, idx

    # -- device ------------------------------------------------------------------------------------
    def context(self):
        ctx = host().mi355h_context(self._h)
        if not ctx:
            raise Mi355Error(host().mi355h_last_error().decode())
        return ctx

    def walk_info(self):
        """(ordered walk available, per-lane stack entries, BVH nodes, boxes in the filtered test's range)"""
        out = (C.c_uint32 * 4)()
        f = lib().mi355i_scene_info
        f.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        _check(f(self.context(), out), "mi355i_scene_info")
        return tuple(int(x) for x in out)

    def traversal_state(self):
        """(scalars[32] uint32, walk, edge, shade) as installed in the device context -- the meaningful part of each stream."""
        f = lib().mi355i_fetch_traversal
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        sc = np.zeros(32, np.uint32)
        _check(f(self.context(), 0, sc.ctypes.data, sc.nbytes), "mi355i_fetch_traversal")
        tri_base, T = int(sc[17]), self.nt
        n4 = tri_base + 2 * T + 4 * (tri_base // 2)     # walk records (2 float4 per inner node) + triangle blocks (2 per triangle) + wide records (4 per inner node)
        out = [sc]
        for which, n in ((1, n4 * 4), (2, T * 12), (3, T * 20)):
            a = np.zeros(n, np.uint32)
            _check(f(self.context(), which, a.ctypes.data, a.nbytes), "mi355i_fetch_traversal")
            out.append(a)
        return tuple(out)

    def build_bvh_device(self):
        """mi355_build_bvh: the reference's tree built on the GPU.  Returns (nodes[n,8] uint32, tri_idx[T] int32, max depth)
        and installs the tree in the device context (the host-side Scene keeps whatever tree it had)."""
        T = self.nt
        nodes = np.zeros((2 * T + 2, 8), np.uint32)
        idx = np.zeros(T, np.int32)
        n, depth = C.c_uint32(0), C.c_int32(0)
        f = lib().mi355_build_bvh
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_int32)]
        _check(f(self.context(), nodes.ctypes.data, idx.ctypes.data, C.byref(n), C.byref(depth)), "mi355_build_bvh")
        return nodes[:n.value].copy(), idx, depth.value

    def set_bvh_arrays(self, nodes: np.ndarray, tri_idx: np.ndarray):
        """mi355_scene_set_bvh with caller-supplied arrays (the reference's 32-byte nodes + triIndexList)."""
        nodes = np.ascontiguousarray(nodes, np.uint32)
        tri_idx = np.ascontiguousarray(tri_idx, np.int32)
        _check(lib().mi355_scene_set_bvh(self.context(), nodes.ctypes.data, nodes.shape[0], tri_idx.ctypes.data,
                                         tri_idx.shape[0]), "mi355_scene_set_bvh")

    def shadowmap_render(self, slot: int, l: Light, size: int = 1024, fetch: bool = False):
        out = np.empty((size, size), np.float32) if fetch else None
        _check(lib().mi355_shadowmap_render(self.context(), slot, C.byref(l), size,
                                            out.ctypes.data if fetch else None), "mi355_shadowmap_render")
        return out

    def light_update(self, slot: int, pos, size: int = 1024, stream: int = 0) -> "Light":
        """mi355_light_update: the slot's shadow map redrawn for a light at `pos`, asynchronously on `stream`; returns the light
        with pos and world_to_light filled in (camera-space members: light() per frame)."""
        l = Light()
        f = lib().mi355_light_update
        f.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_int, C.POINTER(Light), C.c_void_p]
        _check(f(self.context(), slot, (C.c_float * 3)(*[float(x) for x in pos]), size, C.byref(l), C.c_void_p(stream)), "mi355_light_update")
        return l

    def shadowmap_set(self, slot: int, m: np.ndarray):
        m = np.ascontiguousarray(m, np.float32)
        _check(lib().mi355_shadowmap_set(self.context(), slot, m.ctypes.data, m.shape[0]), "mi355_shadowmap_set")

    def render(self, mode: int, cam: Camera, lights, n_lights: int, opts: Opts, want_f32: bool = False,
               pitch_words: int | None = None):
        """One frame through mi355_render().  Returns (xrgb[H,W] uint32, f32[H,W,3] or None, Stats)."""
        W, H = opts.width, opts.height
        rows = H
        if opts.band_count > 1 and opts.compact_rows:
            rows = sum(1 for y in range(H) if (y // opts.band_rows) % opts.band_count == opts.band_index)
        pw = pitch_words or W
        out = np.zeros((rows, pw), np.uint32)
        outf = np.zeros((rows, W, 3), np.float32) if want_f32 else None
        st = Stats()
        _check(lib().mi355_render(self.context(), mode, C.byref(cam), lights, n_lights, C.byref(opts),
                                  out.ctypes.data, pw * 4, outf.ctypes.data if want_f32 else None, C.byref(st)),
               "mi355_render")
        return out[:, :W], outf, st

    def render_into(self, mode: int, cam: Camera, lights, n_lights: int, opts: Opts, out: np.ndarray) -> Stats:
        """mi355_render() into a caller-owned uint32 array (rows x pitch words), e.g. one registered with host_register."""
        st = Stats()
        _check(lib().mi355_render(self.context(), mode, C.byref(cam), lights, n_lights, C.byref(opts), out.ctypes.data,
                                  out.strides[0], None, C.byref(st)), "mi355_render")
        return st

    def render_async(self, mode: int, cam: Camera, lights, n_lights: int, opts: Opts, out: np.ndarray) -> int:
        """mi355_render_async(): enqueue a frame that lands in `out` (uint32, rows x pitch words); returns its ticket."""
        f = lib().mi355_render_async
        f.argtypes = [C.c_void_p, C.c_int, C.POINTER(Camera), C.POINTER(Light), C.c_int, C.POINTER(Opts), C.c_void_p, C.c_int,
                      C.POINTER(C.c_int)]
        t = C.c_int(0)
        _check(f(self.context(), mode, C.byref(cam), lights, n_lights, C.byref(opts), out.ctypes.data, out.strides[0], C.byref(t)),
               "mi355_render_async")
        return t.value

    def render_wait(self, ticket: int) -> Stats:
        f = lib().mi355_render_wait
        f.argtypes = [C.c_void_p, C.c_int, C.POINTER(Stats)]
        st = Stats()
        _check(f(self.context(), ticket, C.byref(st)), "mi355_render_wait")
        return st

    def host_register(self, a: np.ndarray) -> None:
        """Page-lock a caller-owned output array (mi355_host_register).  Use memory with a mapping of its own (own_mapping_array), or
        better host_array(): registering pieces of the malloc heap is what mi355_render.h warns about."""
        f = lib().mi355_host_register
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        _check(f(self.context(), a.ctypes.data, a.nbytes), "mi355_host_register")

    def host_unregister(self, a: np.ndarray) -> None:
        f = lib().mi355_host_unregister
        f.argtypes = [C.c_void_p, C.c_void_p]
        _check(f(self.context(), a.ctypes.data), "mi355_host_unregister")

    def render_device(self, mode: int, cam: Camera, lights, n_lights: int, opts: Opts, d_out: int,
                      pitch_bytes: int, d_outf: int = 0, stream: int = 0):
        """Asynchronous frame into device memory (torch tensor data_ptr) on HIP stream `stream`."""
        _check(lib().mi355_render_device(self.context(), mode, C.byref(cam), lights, n_lights, C.byref(opts),
                                         d_out, pitch_bytes, d_outf or None, stream or None),
               "mi355_render_device")

    def render_batch_device(self, mode: int, cams, lights_per_frame, n_lights: int, opts: Opts, d_outs, pitch_bytes: int,
                            d_outfs=None, stream: int = 0):
        """mi355_render_batch_device: len(cams) frames in one launch.  cams: list of Camera; lights_per_frame: list of
        Light arrays (n_lights used of each); d_outs / d_outfs: device pointers per frame."""
        n = len(cams)
        cam_arr = (Camera * n)(*cams)
        light_arr = (Light * (n * max(n_lights, 1)))()
        for f in range(n):
            for i in range(n_lights):
                light_arr[f * n_lights + i] = lights_per_frame[f][i]
        outs = (C.c_void_p * n)(*[int(p) for p in d_outs])
        outfs = (C.c_void_p * n)(*[int(p) for p in d_outfs]) if d_outfs else None
        f = lib().mi355_render_batch_device
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                      C.c_void_p, C.c_void_p]
        _check(f(self.context(), mode, n, cam_arr, light_arr, n_lights, C.byref(opts), outs, pitch_bytes, outfs, stream or None),
               "mi355_render_batch_device")

    def fetch_stats(self) -> Stats:
        st = Stats()
        _check(lib().mi355_fetch_stats(self.context(), C.byref(st)), "mi355_fetch_stats")
        return st

    def culled_rays(self) -> int:
        """Of the last call's normal_rays, the camera rays that were never generated: pixels of the tiles the tile culling set to
        black without tracing (mi355_stats counts them, like the reference does)."""
        f = lib().mi355i_fetch_culled_rays
        f.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        v = C.c_uint64(0)
        _check(f(self.context(), C.byref(v)), "mi355i_fetch_culled_rays")
        return int(v.value)

    def render_frame_cxx(self, mode: int, width: int, height: int, eye, lookat, light_positions):
        """One frame through the C++ Scene::render* API (mi355::Scene, what a front-end calls)."""
        out = np.zeros((height, width), np.uint32)
        lp = np.ascontiguousarray(np.asarray(light_positions, np.float32).reshape(-1, 3))
        st = Stats()
        rc = host().mi355h_render_frame(self._h, mode, width, height, (C.c_float * 3)(*eye), (C.c_float * 3)(*lookat),
                                        lp.ctypes.data_as(C.POINTER(C.c_float)), lp.shape[0], out.ctypes.data,
                                        C.byref(st))
        if rc != 0:
            raise Mi355Error(host().mi355h_last_error().decode())
        return out, st

    def opts(self) -> Opts:
        return host().mi355h_opts(self._h).contents


def rgb_bytes(xrgb: np.ndarray) -> bytes:
    """Raw R,G,B bytes, row-major from the top row: the layout the reference's frame pins hash."""
    a = np.ascontiguousarray(xrgb)
    return np.stack([(a >> 16) & 0xff, (a >> 8) & 0xff, a & 0xff], axis=-1).astype(np.uint8).tobytes()
