import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session", autouse=True)
def _built_extension():
    """The HIP extension (cross-compiled by hipcc, no GPU needed) must exist before any test that
    loads it -- built artefacts are git-ignored, so a fresh checkout has none."""
    from discregrid_amd.build import build
    build()


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    import dgtest as T

    return np.load(os.path.join(T.GOLDEN, "ref_vectors.npz"))
