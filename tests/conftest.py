import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _ensure_built():
    """The shared objects are git-ignored build products: build them in-tree when a fresh checkout
    runs the tests without having called __graft_entry__.build() first (hipcc cross-compiles)."""
    from raptor_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from raptor_amd import build as rq_build
        rq_build.build()
    from oracle import oracle as O
    O.build()


_ensure_built()


def _has_gpu():
    try:
        import ctypes
        from raptor_amd import _lib
        n = ctypes.c_int()
        _lib.load().rq_device_count(ctypes.byref(n))
        return n.value > 0
    except Exception:
        return False


HAS_GPU = _has_gpu()


def pytest_collection_modifyitems(config, items):
    # a gpu-marked test on a box without a GPU is an error of the invocation, not a skip:
    # `-m gpu` must never pass silently without running the HIP path.
    if HAS_GPU:
        return
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(pytest.mark.skipif(
                not config.getoption("-m") or "gpu" not in config.getoption("-m") or
                "not gpu" in config.getoption("-m"),
                reason="no HIP device in this container"))


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def weights():
    w = np.fromfile(os.path.join(ROOT, "raptor_amd", "data", "raptor_policy.bin"), "<f4")
    assert w.size == 2084
    return w


def _kat(name):
    x = np.fromfile(os.path.join(GOLDEN, f"kat_{name}_input.bin"), "<f4").reshape(500, 2, 22)
    y = np.fromfile(os.path.join(GOLDEN, f"kat_{name}_output.bin"), "<f4").reshape(500, 2, 4)
    return x, y


@pytest.fixture(scope="session", params=["h", "h5"])
def kat(request):
    """The two known-answer vectors the reference ships (checkpoint.h:197-215, checkpoint.h5:/example)."""
    return _kat(request.param)


@pytest.fixture(scope="session")
def device():
    import raptor_amd.l2f as l2f
    return l2f.Device(0)


@pytest.fixture(scope="module")
def w1k(device, oracle):
    """A 1 000-env world on the GPU with its oracle mirror (tests/gpu_common.py), shared by the tests of a module."""
    from gpu_common import World
    return World(device, oracle, 1000)
