import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Make sure librfx.so and the oracle exist (build() is cheap when everything is up to date)."""
    import __graft_entry__ as g
    lib = os.path.join(ROOT, "rayforce_amd", "librfx.so")
    orc = os.path.join(ROOT, "oracle", "librfo.so")
    if not (os.path.exists(lib) and os.path.exists(orc)):
        g.build()
    return True


@pytest.fixture(scope="session")
def eng(built):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from rayforce_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()
