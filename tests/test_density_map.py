"""K3 -- the SPH boundary density map of cmd/generate_density_map: oracle pinning against the
reference's outputs, quadrature constants, and the product's per-node code run on the host
(tests/emu).  GPU parity is in test_gpu_density_map.py."""
import os
import re

import numpy as np
import pytest

import dgtest as T
import emu

DBL_MAX = np.finfo(np.float64).max


def test_gauss_rule_constants():
    """dg_gauss16.h carries the doubles of the reference's table row p = 30 (needed for bit
    parity; they deviate from the exact Gauss-Legendre weights by up to ~1e-13 relative) next to
    correctly rounded exact values computed from scratch."""
    txt = open(os.path.join(T.ROOT, "discregrid_amd", "csrc", "dg_gauss16.h")).read()

    def arr(name):
        body = txt[txt.index("%s[16]" % name):]
        body = body[:body.index("};")]
        return np.array([float.fromhex(v) for v in re.findall(r"-?0x[0-9a-f.]+p[-+]?\d+", body)])
    x, w, xe, we = arr("kGaussX"), arr("kGaussW"), arr("kGaussXExact"), arr("kGaussWExact")
    gx, gw = T.gauss_rule_p30()
    np.testing.assert_array_equal(x, gx)
    np.testing.assert_array_equal(w, gw)
    nx, nw = np.polynomial.legendre.leggauss(16)
    np.testing.assert_allclose(xe, nx, rtol=0, atol=2e-16)
    np.testing.assert_allclose(we, nw, rtol=0, atol=5e-16)
    assert np.abs(w - we).max() < 1e-13 and (np.diff(x) > 0).all()
    if os.path.exists(os.path.join(T.REF_ROOT, "cmd")):
        rx, rw = T.parse_reference_gauss_rule(30)
        np.testing.assert_array_equal(x, rx)
        np.testing.assert_array_equal(w, rw)


@pytest.mark.parametrize("name,res,h,key", [("torus_9_14_6.cdf", [9, 14, 6], 0.15, "torus_density_h015"),
                                             ("torus_16_16_6.cdf", [16, 16, 6], 0.1, "torus16_density_h01")])
def test_oracle_and_product_code_vs_reference(golden, name, res, h, key):
    g = T.read_cdf(os.path.join(T.GOLDEN, name))
    want = golden[key]
    n = len(want)
    # a spread of node ranges (the whole lattice would take a while on few cores)
    rng = np.random.default_rng(5)
    for b in sorted(rng.integers(0, n - 300, size=4).tolist()):
        e = b + 300
        got = T.oracle_density_map(g["domain"], res, g["nodes"][0], h, 1000.0, True, b, e)
        np.testing.assert_array_equal(got, want[b:e])                 # oracle == reference, bit for bit
        got2 = emu.density_map(g["domain"], res, g["nodes"][0], h, 1000.0, True, b, e)
        np.testing.assert_array_equal(got2, want[b:e])                # product per-node code == reference
    assert (want == DBL_MAX).any() or name.startswith("torus_9")


@pytest.mark.parametrize("tag,h", [("h012", 0.12), ("h045", 0.45)])
def test_box_sized_grid_vs_reference(tag, h):
    """tests/golden/density_box.npz (torus 12 x 11 x 9, small and large support radius, made by the
    unmodified reference): the oracle is pinned on it here, the GPU kernel in test_gpu_density_map.py."""
    d = np.load(os.path.join(T.GOLDEN, "density_box.npz"))
    want = d["density_" + tag]
    integrated = (want != DBL_MAX) & (want != 0.0)
    assert integrated.sum() > 500
    n = len(want)
    for b in (0, n // 3, n - 400):
        got = T.oracle_density_map(d["domain"], d["res"], d["sdf"], h, 1000.0, True, b, b + 400)
        np.testing.assert_array_equal(got, want[b:b + 400])


def test_no_predicate_and_table_mode(golden):
    g = T.read_cdf(os.path.join(T.GOLDEN, "torus_9_14_6.cdf"))
    want = golden["torus_density_h015_nopred"]
    b, e = 2000, 2200
    np.testing.assert_array_equal(T.oracle_density_map(g["domain"], [9, 14, 6], g["nodes"][0], 0.15, 1000.0, False, b, e),
                                  want[b:e])
    got = emu.density_map(g["domain"], [9, 14, 6], g["nodes"][0], 0.15, 1000.0, False, b, e, cells=g["cells"][0],
                          cell_map=g["cell_map"][0])
    np.testing.assert_array_equal(got, want[b:e])


@pytest.mark.skipif(not T.ref_available(), reason="oracle/_ref not built")
def test_against_live_reference():
    V, F = T.icosphere(6)
    dom = T.ref_default_domain(V)
    res = [8, 7, 9]
    g = T.RefGrid(V, F, dom, res)
    g.add_sdf()
    sdf = g.nodes(0)
    g.add_density_map(0.2, 1000.0)
    want = g.nodes(1)
    np.testing.assert_array_equal(T.oracle_density_map(dom, res, sdf, 0.2, 1000.0, True), want)
    np.testing.assert_array_equal(emu.density_map(dom, res, sdf, 0.2, 1000.0, True), want)
    assert ((want != DBL_MAX) & (want != 0)).sum() > 100
