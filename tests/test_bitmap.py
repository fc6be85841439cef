"""DiscreteFieldToBitmap (reference cmd/discrete_field_to_bitmap): the CPU
restatement (oracle/bitmap_slice.py + the oracle's interpolate) reproduces the bitmaps the
unmodified reference tool wrote, byte for byte; the GPU tool is compared with the same fixtures
in test_gpu_host_api.py."""
import os
import sys
import numpy as np
import pytest

import dgtest as T

sys.path.insert(0, os.path.join(T.ROOT, "oracle"))
sys.path.insert(0, T.GOLDEN)
import bitmap_slice as B  # noqa: E402
from make_golden_bitmaps import CASES  # noqa: E402


def _opts(opts):
    o = {"-f": "0", "-s": "1024", "-p": "xy", "-d": "0", "-c": "gb"}
    o.update(dict(zip(opts[::2], opts[1::2])))
    return int(o["-f"]), int(o["-s"]), o["-p"], float(o["-d"]), o["-c"]


@pytest.mark.parametrize("name", sorted(CASES))
def test_restatement_matches_reference_bitmap(name):
    src, opts = CASES[name]
    f, s, plane, depth, cm = _opts(opts)
    g = T.read_cdf(os.path.join(T.GOLDEN, src))
    P, w, h = B.slice_points(g["domain"], plane, depth, s)
    phi = T.oracle_interpolate(g["domain"], g["res"], g["nodes"][f], P, cells=g["cells"][f], cell_map=g["cell_map"][f])
    got = B.bmp_bytes(B.colour(phi, cm), w, h)
    want = open(os.path.join(T.GOLDEN, "bitmap_%s.bmp" % name), "rb").read()
    assert len(got) == len(want)
    diff = np.flatnonzero(np.frombuffer(got, np.uint8) != np.frombuffer(want, np.uint8))
    assert diff.size == 0, "first differing byte at offset %d" % diff[0]


def test_plane_axes_follow_the_reference_rules():
    assert B.plane_axes("xy") == [0, 1, 2]
    assert B.plane_axes("xz") == [0, 2, 1]
    assert B.plane_axes("yz") == [1, 2, 0]
    assert B.plane_axes("yx") == [1, 0, 2]
    assert B.plane_axes("zx") == [2, 0, 1]
    assert B.plane_axes("zy") == [2, 1, 0]


def test_fixture_images_are_not_trivial():
    for name in CASES:
        b = np.frombuffer(open(os.path.join(T.GOLDEN, "bitmap_%s.bmp" % name), "rb").read(), np.uint8)[54:]
        assert len(np.unique(b)) > 8, name
