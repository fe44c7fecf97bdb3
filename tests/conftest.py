import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`pytest tests/` on a box without a HIP device skips the GPU tests instead of failing them (a missing or broken native
    library is no excuse: then they run and fail)."""
    reason = None
    try:
        import renderer_amd
        if renderer_amd.device_count() < 1:
            reason = "no HIP device on this box"
    except Exception:
        pass          # the native library is missing or does not load: NOT a reason to skip -- the GPU tests fail loudly
    if reason:
        skip = pytest.mark.skip(reason=reason)
        for it in items:
            if "gpu" in it.keywords:
                it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_ctypes as O
    O.build()
    return O


_SCENES = {}


@pytest.fixture(scope="session")
def oracle_scene(oracle):
    """Oracle scenes (with BVH) cached per session; BVH cached on disk in the scratch dir."""
    from renderer_amd import assets

    def get(name, bvh=False):
        key = name
        if key not in _SCENES:
            _SCENES[key] = oracle.Scene(assets.oracle_path(name))
        s = _SCENES[key]
        if bvh and s.num_nodes == 0:
            s.bvh_ensure(os.path.join(assets.cache_dir(), name + ".oracle.bvh"))
        return s
    return get


@pytest.fixture(autouse=True)
def _host_trace_marker(request):
    """MI355_HOST_TRACE=<file> (capi.hip: host_trace): name the test in the library's trace of GPU writes into host memory."""
    path = os.environ.get("MI355_HOST_TRACE")
    if path:
        with open(path, "a") as f:
            f.write("TEST %s\n" % request.node.nodeid)
    yield
