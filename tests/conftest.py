import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_sessionstart(session):
    # Tests that hand torch-allocated device buffers to the C ABI need torch's bundled HIP runtime
    # to be the one the process loads first: libhiptrace.so then binds to the same runtime by
    # SONAME.  Loaded the other way round, torch no longer sees the GPU.
    try:
        import torch
        torch.cuda.is_available()
    except Exception:
        pass


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs oracle/_ref/libref.so built from the reference sources")


@pytest.fixture(scope="session")
def host():
    from raytracer_amd import host as h
    return h


@pytest.fixture(scope="session")
def oracle():
    import oracle as o
    return o
