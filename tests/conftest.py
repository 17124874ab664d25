import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_ctx():
    """A Context on cuda:0 / HIP device 0.  Fails loudly (no skip, no fallback) if the HIP
    library is missing or no device is usable."""
    from swarm_amd import Context
    ctx = Context(0)
    yield ctx
    ctx.close()
