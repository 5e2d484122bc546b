import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
