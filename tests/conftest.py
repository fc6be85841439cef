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


def k1_variant_fixture():
    """Autouse fixture factory for the GPU modules that exercise K1 / K1p on SMALL meshes: every test runs
    with the exact kernel only (DG_FORCE=k1_fast=0) and with the filtered kernel forced (k1_fast=1; by default
    meshes below dg::kFastMinTriangles triangles keep the exact kernel, so without this the filtered path
    -- in-wave exact fallback, seed parking, degenerate triangles, out-of-range points -- would only be
    covered by the large digest configs)."""
    @pytest.fixture(autouse=True, params=["exact", "filtered"])
    def k1_variant(request, monkeypatch):
        import dgtest as T
        T.force(monkeypatch, k1_fast="1" if request.param == "filtered" else "0")
        return request.param
    return k1_variant
