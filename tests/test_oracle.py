"""Pins the CPU restatement (oracle/discregrid_oracle.cpp) -- the checker every GPU parity
test relies on -- against (1) the reference's only golden file box.cdf, (2) golden vectors
generated from the unmodified reference (tests/golden/make_golden.py) and (3) the
unmodified reference itself whenever oracle/_ref is built.  CPU only."""
import os

import numpy as np
import pytest

import dgtest as T

DBL_MAX = np.finfo(np.float64).max
MESHES = {"box": T.box_mesh, "ico8": lambda: T.icosphere(8), "torus": T.torus, "bunny": T.bunny_mesh}


def test_box_cdf_known_answer(tmp_path):
    """GenerateSDF -r "5 5 5" box.obj == cmd/generate_sdf/resources/box.cdf, byte for byte
    (default domain rule, addFunction, signed_distance, cell table, save format)."""
    V, F = T.box_mesh()
    dom = T.oracle_default_domain(V)
    np.testing.assert_array_equal(dom[:3], -1.0034701016151377)
    np.testing.assert_array_equal(dom[3:], 1.0034641016151378)
    coeffs = T.OracleMesh(V, F).sample_nodes(dom, [5, 5, 5])
    assert len(coeffs) == 1296
    p = str(tmp_path / "box.cdf")
    assert T.oracle_write_cdf(p, dom, [5, 5, 5], [coeffs]) == 27040
    with open(p, "rb") as a, open(os.path.join(T.GOLDEN, "box.cdf"), "rb") as b:
        assert a.read() == b.read()


def test_icosphere_shape():
    V, F = T.icosphere(71)
    assert V.shape == (50412, 3) and F.shape == (100820, 3)
    np.testing.assert_allclose(np.linalg.norm(V, axis=1), 1.0, atol=1e-15)
    d = T.OracleMesh(V, F).signed_distance(np.array([[0.0, 0, 0], [1.2, 1.2, 1.2]]))
    assert -1.0 < d[0] < -0.9999 and abs(d[1] - 1.0785) < 1e-3


@pytest.mark.parametrize("name", list(MESHES))
def test_sdf_coefficients_golden(golden, name):
    V, F = MESHES[name]()
    dom, res = golden[name + "_domain"], golden[name + "_res"]
    np.testing.assert_array_equal(T.oracle_default_domain(V), dom)
    got = T.OracleMesh(V, F).sample_nodes(dom, res)
    np.testing.assert_array_equal(got, golden[name + "_coeffs"])  # bit-exact


@pytest.mark.parametrize("name", list(MESHES))
def test_signed_distance_golden(golden, name):
    V, F = MESHES[name]()
    d, tri, ent, near = T.OracleMesh(V, F).signed_distance(golden[name + "_P"], full=True)
    np.testing.assert_array_equal(d, golden[name + "_sd"])
    np.testing.assert_array_equal(tri, golden[name + "_tri"])
    np.testing.assert_array_equal(ent, golden[name + "_ent"])
    np.testing.assert_array_equal(near, golden[name + "_near"])


@pytest.mark.parametrize("name", list(MESHES))
def test_interpolate_golden(golden, name):
    dom, res = golden[name + "_domain"], golden[name + "_res"]
    phi, grad = T.oracle_interpolate(dom, res, golden[name + "_coeffs"], golden[name + "_P"], grad=True)
    inside = golden[name + "_phi"] != DBL_MAX
    assert inside.any() and (~inside).any()
    np.testing.assert_array_equal(phi, golden[name + "_phi"])
    np.testing.assert_array_equal(grad[inside], golden[name + "_grad"][inside])
    np.testing.assert_array_equal(T.oracle_interpolate(dom, res, golden[name + "_coeffs"], golden[name + "_P"]),
                                  golden[name + "_phi"])


@pytest.mark.parametrize("name", list(MESHES))
def test_node_positions_golden(golden, name):
    dom, res = golden[name + "_domain"], golden[name + "_res"]
    pos = T.oracle_node_positions(dom, res)
    np.testing.assert_array_equal(pos[golden[name + "_pos_idx"]], golden[name + "_pos"])


def test_inverted_sdf_golden(golden):
    V, F = T.torus()
    got = T.OracleMesh(V, F).sample_nodes(T.oracle_default_domain(V), [7, 7, 7], invert=True)
    np.testing.assert_array_equal(got, golden["torus_inv_coeffs"])


def test_config3_and_config2_lattice_samples(golden):
    """Strided samples of the judged lattices: icosphere nu=71 at 256^3, bunny at 128^3."""
    for key, mesh, res in (("ico71", lambda: T.icosphere(71), [256] * 3), ("bunny128", T.bunny_mesh, [128] * 3)):
        V, F = mesh()
        om = T.OracleMesh(V, F)
        dom = golden[key + "_domain"]
        np.testing.assert_array_equal(T.oracle_default_domain(V), dom)
        idx = golden[key + "_lattice_idx"]
        pos = np.stack([T.oracle_node_positions(dom, res, int(l), int(l) + 1)[0] for l in idx])
        np.testing.assert_array_equal(om.signed_distance(pos), golden[key + "_lattice_sd"])
    np.testing.assert_array_equal(om.__class__(*T.icosphere(71)).signed_distance(golden["ico71_P"]),
                                  golden["ico71_sd"])


def test_shape_function_properties():
    """Partition of unity, zero-sum derivatives, nodal interpolation, FD gradient (the
    commented-out recipe at cubic_lagrange_discrete_grid.cpp:1028-1042, eps = 1e-6)."""
    rng = np.random.default_rng(7)
    xi = rng.uniform(-1, 1, size=(256, 3))
    N, dN = T.oracle_shape(xi, grad=True)
    np.testing.assert_allclose(N.sum(axis=1), 1.0, atol=1e-14)
    np.testing.assert_allclose(dN.sum(axis=1), 0.0, atol=1e-13)
    eps = 1e-6
    for d in range(3):
        e = np.zeros(3)
        e[d] = eps
        fd = (T.oracle_shape(xi + e) - T.oracle_shape(xi - e)) / (2 * eps)
        np.testing.assert_allclose(fd, dN[:, :, d], atol=5e-9)
    # nodal property at the 8 corners
    corners = np.array([[sx, sy, sz] for sz in (-1, 1) for sy in (-1, 1) for sx in (-1, 1)], dtype=np.float64)
    Nc = T.oracle_shape(corners)
    np.testing.assert_allclose(Nc[:, :8], np.eye(8), atol=1e-15)
    np.testing.assert_allclose(Nc[:, 8:], 0.0, atol=1e-15)


def test_interpolation_reproduces_cubic_polynomial():
    """The serendipity-cubic space contains all polynomials of total degree <= 3."""
    dom = np.array([-1.0, -0.5, 0.0, 1.5, 0.75, 2.0])
    res = [3, 4, 2]

    def f(p):
        x, y, z = p[:, 0], p[:, 1], p[:, 2]
        return 1 + x - 2 * y + 0.5 * z + x * y - y * z + x * x * z - 0.3 * y ** 3 + 0.7 * x * y * z

    coeffs = f(T.oracle_node_positions(dom, res))
    P = np.random.default_rng(3).uniform(dom[:3], dom[3:], size=(500, 3))
    np.testing.assert_allclose(T.oracle_interpolate(dom, res, coeffs, P), f(P), atol=1e-12)


def test_analytic_sphere_and_box_sdf():
    V, F = T.icosphere(16)
    P = np.random.default_rng(5).uniform(-1.5, 1.5, size=(2000, 3))
    d = T.OracleMesh(V, F).signed_distance(P)
    np.testing.assert_allclose(d, np.linalg.norm(P, axis=1) - 1.0, atol=6e-3)
    V, F = T.box_mesh()
    d = T.OracleMesh(V, F).signed_distance(P)
    q = np.abs(P) - 1.0
    exact = np.linalg.norm(np.maximum(q, 0), axis=1) + np.minimum(q.max(axis=1), 0)
    np.testing.assert_allclose(d, exact, atol=1e-12)


def test_cell_table_closed_form_consistency():
    res = [3, 2, 4]
    cells = T.oracle_cell_table(res)
    dom = np.array([0, 0, 0, 3.0, 2.0, 4.0])
    pos = T.oracle_node_positions(dom, res)
    # node j of each cell sits at the abscissae of cubic_lagrange_discrete_grid.cpp:58-94
    t = 1.0 / 3.0
    absc = [[sx, sy, sz] for sz in (-1, 1) for sy in (-1, 1) for sx in (-1, 1)]
    absc += [[s, sy, sz] for sy in (-1, 1) for sz in (-1, 1) for s in (-t, t)]
    absc += [[sx, s, sz] for sz in (-1, 1) for sx in (-1, 1) for s in (-t, t)]
    absc += [[sx, sy, s] for sx in (-1, 1) for sy in (-1, 1) for s in (-t, t)]
    absc = np.array(absc)
    for c in range(len(cells)):
        k, r = divmod(c, res[0] * res[1])
        j, i = divmod(r, res[0])
        lo = np.array([i, j, k], dtype=np.float64)
        expect = lo + (absc + 1) / 2
        np.testing.assert_allclose(pos[cells[c]], expect, atol=1e-12)


@pytest.mark.skipif(not T.ref_available(), reason="oracle/_ref not built (needs /root/reference)")
@pytest.mark.parametrize("name", ["box", "ico8", "torus", "bunny"])
def test_against_live_reference(name):
    """Bit-for-bit against the unmodified reference on inputs that are NOT in the goldens."""
    V, F = MESHES[name]()
    dom = T.ref_default_domain(V)
    res = [7, 9, 8]
    g = T.RefGrid(V, F, dom, res)
    g.add_sdf()
    om = T.OracleMesh(V, F)
    coeffs = om.sample_nodes(dom, res)
    np.testing.assert_array_equal(coeffs, g.nodes())
    np.testing.assert_array_equal(T.oracle_cell_table(res), g.cells())
    c_ref, c_or = g.construction(), om.construction()
    inner = c_ref["children"][:, 0] != -1
    for k in ("pn_tri", "pn_edge", "pn_vert", "children"):
        np.testing.assert_array_equal(c_ref[k], c_or[k])
    np.testing.assert_array_equal(c_ref["spheres"][inner], c_or["spheres"][inner])
    P = np.random.default_rng(99).uniform(dom[:3], dom[3:], size=(3000, 3))
    for a, b in zip(g.signed_distance(P, full=True), om.signed_distance(P, full=True)):
        np.testing.assert_array_equal(a, b)
    a, ga = g.interpolate(P, grad=True)
    b, gb = T.oracle_interpolate(dom, res, coeffs, P, grad=True)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(ga, gb)


def test_lattice_digests_pin_the_oracle():
    """tests/golden/lattice_digests.npz (SHA-256 per 2^20-node block of the REFERENCE's coefficients for the
    judged configurations, make_digests.py) against the oracle on a few blocks: heavy-brick region of the
    256^3 icosphere lattice, a bunny block, the last (ragged) block of the 512^3 lattice."""
    import os
    gold = dict(np.load(os.path.join(T.GOLDEN, "lattice_digests.npz")))
    B = int(gold["block"])
    assert B == 1 << 20
    for name, mesh, res, blocks in (("bunny128", T.bunny_mesh(), [128] * 3, [3]),
                                    ("ico71_256", T.icosphere(71), [256] * 3, [8]),
                                    ("ico71_512", T.icosphere(71), [512] * 3, [-1])):
        V, F = mesh
        dom = gold[name + "_domain"]
        np.testing.assert_array_equal(dom, T.oracle_default_domain(V))
        n = T.n_nodes(res)
        assert n == int(gold[name + "_nodes"]) and len(gold[name + "_digest"]) == (n + B - 1) // B
        om = T.OracleMesh(V, F)
        for b in blocks:
            b = b % len(gold[name + "_digest"])
            e = min(n, (b + 1) * B)
            got = T.block_digests(om.sample_nodes(dom, res, b * B, e), B)
            np.testing.assert_array_equal(got[0], gold[name + "_digest"][b])
    # config 5 keys are present and sized for 10 M queries
    nq = int(gold["c5_n_queries"])
    for k in ("c5_uniform_phi", "c5_uniform_phi_g", "c5_uniform_grad", "c5_shell_points", "c5_shell_phi", "c5_shell_phi_g", "c5_shell_grad"):
        assert len(gold[k]) == (nq + B - 1) // B
