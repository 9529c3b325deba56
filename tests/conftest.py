import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _assets():
    """Scene assets are generated, not committed (rendering_amd/assets.py)."""
    from rendering_amd import assets
    assets.ensure()
    os.makedirs(os.path.join(ROOT, "output"), exist_ok=True)
    os.chdir(ROOT)
    yield


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def ra():
    import rendering_amd as RA
    RA.load()
    return RA
