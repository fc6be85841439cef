"""A short run of the randomised parity campaign (tests/perf/fuzz_parity.py): random meshes (ellipsoids,
tori, spiky blobs, triangle soups, scaled boxes, a bunny with holes) x random lattices and domains x
random points, K1 / K1p / K2 against the oracle, bit for bit."""
import os
import subprocess
import sys

import pytest

import dgtest as T

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed,k1_fast", [(11, "0"), (12, "1"), (13, "1")])
def test_fuzz_parity_short(seed, k1_fast):
    """(k1_fast: the exact K1 kernel only / the filtered kernel forced also for the small random meshes)"""
    script = os.path.join(T.ROOT, "tests", "perf", "fuzz_parity.py")
    out = subprocess.run([sys.executable, script, "6", str(seed)], capture_output=True, text=True, timeout=300,
                         env=T.force_env(k1_fast=k1_fast))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "fuzz ok" in out.stdout
