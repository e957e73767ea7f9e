import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle32():
    from oracle.raster_oracle import RasterOracle
    import numpy as np
    return RasterOracle(np.float32)


@pytest.fixture(scope="session")
def oracle64():
    from oracle.raster_oracle import RasterOracle
    import numpy as np
    return RasterOracle(np.float64)


@pytest.fixture
def host_density():
    """Opt in to the HOST statement of the factorised density on CPU tensors (table-building code that the CPU tests pin
    against the reference's outputs); without it a likelihood call on a CPU tensor raises (no CPU fallback)."""
    from contextgs_amd import entropy_bottleneck as eb
    eb.ALLOW_HOST_FORWARD = True
    yield
    eb.ALLOW_HOST_FORWARD = False
