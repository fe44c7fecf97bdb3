"""Mesh assets shipped with the repo.

The three meshes BASELINE.json's configs name (chessboard.tri, statue.ply,
dragon_vis.ply -- the reference's own 3D-Objects/ data files) are kept
xz-compressed under ``assets/`` so that they travel to the GPU box, and are
unpacked on first use into a scratch directory (the loaders, like the
reference's, read plain files; a ``<mesh>.bvh`` cache may be written beside
them, Raytracer.cc:747-753).
"""
from __future__ import annotations

import hashlib
import lzma
import os
import tempfile

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSET_DIR = os.path.join(_ROOT, "assets")

# sha256 of the unpacked files (== the reference's 3D-Objects/ files)
SHA256 = {
    "chessboard.tri": "83fc4d47e47ab93526d671eed00a4f7e6115f423a450c98b6e420205a09c5386",
    "statue.ply": "ca4906aeaef5646f69612a41fbab21922cb7512162f571ef65e220dd27a3af9b",
    "dragon_vis.ply": "4d70bb53c2fe06df59d8d8c8087196446919ccc3324c9f286bfcdf0f181027fb",
    "legocar.3ds": "407311b55df3d390a3f66348d06966cd48fb8f249170ca122a83b1f0a65033b9",
    "trainColor.tri": "00e51bd47174c5132a68963dd008d14af78e46f633f403cde52b8449d24485fa",      # the mesh of the reference's `make bench` (src/Makefile.am:25-26)
}

# Test fixtures (tests/golden/): what the REAL lib3ds hands the reference's loader for a .3ds asset, dumped by
# oracle/ref3ds/dump3ds.c (scripts/make_3ds_golden.sh).  The oracle loads this instead of parsing .3ds itself.
GOLDEN_DIR = os.path.join(_ROOT, "tests", "golden")
R3DS_OF = {"legocar.3ds": ("legocar_3ds.r3ds", "4f6414dff890a6923d6182f3e6ccb846b8902ff5b0c6f0ca3ec30bf3e8de0de7")}


def cache_dir() -> str:
    """Private scratch directory of this user (mode 0700, owned by us: a directory somebody else planted is refused)."""
    d = os.environ.get("RENDERER_AMD_CACHE") or os.path.join(
        tempfile.gettempdir(), "renderer_amd_cache_%d" % os.getuid())
    os.makedirs(d, mode=0o700, exist_ok=True)
    st = os.stat(d)
    if st.st_uid != os.getuid() or (st.st_mode & 0o022):
        raise RuntimeError("cache directory %s is not a private directory of uid %d (owner %d, mode %o): set "
                           "RENDERER_AMD_CACHE to one" % (d, os.getuid(), st.st_uid, st.st_mode & 0o777))
    return d


_VERIFIED = set()


def _intact(path: str, sha: str) -> bool:
    """An unpacked file is trusted once per process, after its hash has been checked (a partial file left by a crash or
    a planted one is unpacked again)."""
    if path in _VERIFIED:
        return True
    try:
        with open(path, "rb") as f:
            ok = hashlib.sha256(f.read()).hexdigest() == sha
    except OSError:
        ok = False
    if ok:
        _VERIFIED.add(path)
    return ok


def _unpack(src_xz: str, dst: str, sha: str) -> str:
    if not _intact(dst, sha):
        with lzma.open(src_xz, "rb") as f:
            data = f.read()
        if hashlib.sha256(data).hexdigest() != sha:
            raise RuntimeError("%s is corrupt" % src_xz)
        tmp = dst + ".tmp%d" % os.getpid()
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, dst)
    return dst


def oracle_path(name: str) -> str:
    """What the ORACLE loads for mesh ``name`` (tests only): the mesh itself, or for a .3ds asset the dump of what
    the real lib3ds reads out of it."""
    if name in R3DS_OF:
        fn, sha = R3DS_OF[name]
        return _unpack(os.path.join(GOLDEN_DIR, fn + ".xz"), os.path.join(cache_dir(), fn), sha)
    return mesh_path(name)


def mesh_path(name: str) -> str:
    """Return the path of the unpacked mesh ``name`` (unpacking it if needed)."""
    if name not in SHA256:
        raise KeyError("unknown mesh asset %r (have: %s)" % (name, ", ".join(SHA256)))
    dst = os.path.join(cache_dir(), name)
    if not _intact(dst, SHA256[name]):
        with lzma.open(os.path.join(ASSET_DIR, name + ".xz"), "rb") as f:
            data = f.read()
        if hashlib.sha256(data).hexdigest() != SHA256[name]:
            raise RuntimeError("asset %s is corrupt" % name)
        tmp = dst + ".tmp%d" % os.getpid()
        with open(tmp, "wb") as f:
            f.write(data)
        os.replace(tmp, dst)
    return dst
