import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    """GPU tests run with pytest's fd-level capture suspended: native code (the HIP runtime, glibc) writes its last words to fd 2, and
    under capture they die with the process -- round 5 had an abort inside the library whose message nobody ever saw."""
    capman = item.config.pluginmanager.getplugin("capturemanager")
    if item.get_closest_marker("gpu") is None or capman is None:
        yield
        return
    capman.suspend_global_capture(in_=False)
    try:
        yield
    finally:
        capman.resume_global_capture()


@pytest.fixture(scope="session")
def pkg():
    """The product package (ctypes layer over librtmi.so); builds the library if it is missing."""
    import rtmi_loader
    p = rtmi_loader.load()
    p.build_library()
    return p


@pytest.fixture(scope="session")
def ob():
    """The CPU oracle (test infrastructure)."""
    from oracle import binding
    binding.build()
    return binding


@pytest.fixture(scope="session")
def rtow(ob):
    """S-RTOW(seed 12345): 488 spheres."""
    return ob.make_world_spheres(12345)


GOLDEN = os.path.join(ROOT, "tests", "golden")
