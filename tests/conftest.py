import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def vkr():
    return importlib.import_module("vk-renderer_amd")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("vk-renderer_amd.synth")


@pytest.fixture(scope="session")
def oracle():
    import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def ctx(vkr):
    """One context for the whole GPU session (fails loudly without a gfx950 device)."""
    c = vkr.Context(0)
    yield c
    c.wait_idle()
