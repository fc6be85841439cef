"""A short run of the randomised parity campaign (tests/perf/fuzz_parity.py): random meshes (ellipsoids,
tori, spiky blobs, triangle soups, scaled boxes, a bunny with holes) x random lattices and domains x
random points, K1 / K1p / K2 against the oracle, bit for bit."""
import os
import subprocess
import sys

import pytest

import dgtest as T

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_parity_short(seed):
    script = os.path.join(T.ROOT, "tests", "perf", "fuzz_parity.py")
    out = subprocess.run([sys.executable, script, "6", str(seed)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "fuzz ok" in out.stdout
