"""Edge cases on the GPU through the C ABI: unusual meshes, extreme grid shapes, empty inputs, and
the maximum lattice the reference's 32-bit node index admits for the judged configs (512^3,
943 460 865 nodes, 7.5 GB) on one device."""
import numpy as np
import pytest

import dgtest as T
from test_edge_cases import meshes

from conftest import k1_variant_fixture

pytestmark = pytest.mark.gpu
DBL_MAX = np.finfo(np.float64).max
k1_variant = k1_variant_fixture()


@pytest.fixture(scope="module")
def dg():
    import discregrid_amd
    discregrid_amd.load_library()
    assert discregrid_amd.device_count() >= 1
    return discregrid_amd


@pytest.mark.parametrize("name", list(meshes()))
def test_unusual_meshes(dg, name):
    V, F = meshes()[name]
    om, m = T.OracleMesh(V, F), dg.Mesh(V, F)
    lo, hi = V.min(axis=0), V.max(axis=0)
    ext = max((hi - lo).max(), 1e-3)
    P = np.random.default_rng(17).uniform(lo - ext, hi + ext, size=(3000, 3))
    a, b = m.signed_distance(P), om.signed_distance(P)
    # full equality, sign included, on open / soup / degenerate / non-manifold meshes too (see tests/test_edge_cases.py)
    np.testing.assert_array_equal(a, b)
    dom = np.concatenate([lo - 0.1 * ext, hi + 0.1 * ext])
    got = m.sample_nodes(dg.grid_desc(dom[:3], dom[3:], [5, 4, 3]))
    np.testing.assert_array_equal(got, om.sample_nodes(dom, [5, 4, 3]))
    assert m.info()["not_watertight"] == (0 if name in ("needle_box",) else m.info()["not_watertight"])


def test_pooled_epilogue_of_the_filtered_kernel_both_branches(dg, monkeypatch):
    """k_sample_fast's epilogue on the device (dg_kernels_k1.hip): a valence-40 apex under a dense lattice, the filtered kernel forced
    (80 triangles are below its default threshold), the test counters on (DG_FORCE=pool_stats=1): waves pool the tails of their lists
    AND waves whose tails do not fit the pool run them lane by lane in the same launch (64 lanes x up to nine candidates against 352
    pairs); a pool capped at 8 pairs / at none moves the waves to the second branch.  Every variant: the bits of the oracle, sign
    included.  (The emulator's model of the same arithmetic: tests/test_emu.py, same name.)"""
    V, F = T.bipyramid(40)
    apex = V[40]
    dom = np.concatenate([apex - 0.02, apex + 0.02])
    res = [24, 24, 24]
    want = T.OracleMesh(V, F).sample_nodes(dom, res)
    m = dg.Mesh(V, F)
    g = dg.grid_desc(dom[:3], dom[3:], res)
    import torch
    n = dg.n_nodes(g)
    out = torch.empty(n, dtype=torch.float64, device="cuda")

    def launch():                 # ONE launch over the whole lattice (the host entry point samples in chunks)
        out.fill_(float("nan"))
        m.sample_nodes_device(g, 0, n, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return out.cpu().numpy()
    seen = {}
    for cap in (None, 8, 0):
        T.force(monkeypatch, k1_fast=1, pool_stats=1, pool_cap=cap)
        np.testing.assert_array_equal(launch(), want, err_msg="pool_cap=%s" % cap)
        seen[cap] = m.last_epilogue_stats()
        np.testing.assert_array_equal(m.sample_nodes(g), want, err_msg="pool_cap=%s, host entry point" % cap)
    pooled, unpooled = seen[None]
    assert pooled > 0 and unpooled > 0, seen
    assert seen[8][1] > unpooled and seen[0][0] == 0 and seen[0][1] == pooled + unpooled, seen
    T.force(monkeypatch, pool_stats=None, pool_cap=None)
    np.testing.assert_array_equal(launch(), want)
    assert m.last_epilogue_stats() == (0, 0)          # not counted in production
    # the same lattice through the emulator's model: the same waves take the same branch
    import emu
    try:
        emu.set_pool_cap()
        emu.set_fast(1)
        np.testing.assert_array_equal(emu.EmuMesh(V, F).sample_range(dom, res), want)
        assert emu.pool_stats() == seen[None], (emu.pool_stats(), seen[None])
    finally:
        emu.set_fast(1)


@pytest.mark.parametrize("res", [[1, 1, 1], [1, 7, 2], [33, 1, 1], [2, 2, 64], [3, 5, 4]])
def test_extreme_grid_shapes(dg, res):
    V, F = T.torus()
    dom = T.oracle_default_domain(V)
    g = dg.grid_desc(dom[:3], dom[3:], res)
    want = T.OracleMesh(V, F).sample_nodes(dom, res)
    np.testing.assert_array_equal(dg.Mesh(V, F).sample_nodes(g), want)
    P = np.random.default_rng(1).uniform(dom[:3], dom[3:], size=(500, 3))
    np.testing.assert_array_equal(dg.Field(g, want).interpolate(P), T.oracle_interpolate(dom, res, want, P))


def test_domains_away_from_the_mesh_and_scaled_meshes(dg):
    V, F = T.icosphere(6)
    om, m = T.OracleMesh(V, F), dg.Mesh(V, F)
    for dom in ([5, 5, 5, 6, 6.5, 7], [-0.2, -0.1, -0.3, 0.1, 0.2, 0.0], [-1e3, -1e3, -1e3, 1e3, 1e3, 1e3]):
        dom = np.array(dom, dtype=np.float64)
        np.testing.assert_array_equal(m.sample_nodes(dg.grid_desc(dom[:3], dom[3:], [4, 5, 3])), om.sample_nodes(dom, [4, 5, 3]))
    W = T.icosphere(5)[0] * np.array([30.0, 0.02, 4.0]) + np.array([-500.0, 7.0, -0.001])
    F5 = T.icosphere(5)[1]
    dom = T.oracle_default_domain(W)
    np.testing.assert_array_equal(dg.Mesh(W, F5).sample_nodes(dg.grid_desc(dom[:3], dom[3:], [6, 9, 4])),
                                  T.OracleMesh(W, F5).sample_nodes(dom, [6, 9, 4]))


def test_empty_and_invalid_inputs(dg):
    V, F = T.box_mesh()
    m = dg.Mesh(V, F)
    assert len(m.signed_distance(np.empty((0, 3)))) == 0
    g = dg.grid_desc([0] * 3, [1] * 3, [2, 2, 2])
    f = dg.Field(g, np.zeros(dg.n_nodes(g)))
    assert len(f.interpolate(np.empty((0, 3)))) == 0
    with pytest.raises(dg.DiscregridError):
        dg.Mesh(V, np.empty((0, 3), dtype=np.uint32))                   # empty triangle list
    with pytest.raises(dg.DiscregridError):
        dg.Mesh(V, np.array([[0, 1, 99]], dtype=np.uint32))             # vertex index out of range
    with pytest.raises(dg.DiscregridError):
        dg.Field(g, np.zeros(5))                                        # wrong coefficient count
    with pytest.raises(dg.DiscregridError):
        f.density_map_nodes(dg.n_nodes(g), -1.0, 1000.0)                # non-positive support radius


def test_maximum_lattice_512_cubed(dg):
    """BASELINE config 4's lattice on ONE GPU: 943 460 865 nodes = 7.5 GB of coefficients.  Checked
    on a strided device-side sample against the oracle, and shard 3 of 8 against the same nodes."""
    import torch
    V, F = T.icosphere(71)
    dom = T.oracle_default_domain(V)
    res = [512] * 3
    g = dg.grid_desc(dom[:3], dom[3:], res)
    n = dg.n_nodes(g)
    assert n == 943460865
    m = dg.Mesh(V, F)
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    m.sample_nodes_device(g, 0, n, out.data_ptr(), stream=s)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print("icosphere 512^3 on one GPU: %.1f ms, %.1f Mnodes/s" % (ms, n / ms / 1e3))
    idx = np.arange(7, n, 1000003, dtype=np.int64)
    got = out[torch.from_numpy(idx).cuda()].cpu().numpy()
    pos = np.stack([T.oracle_node_positions(dom, res, int(l), int(l) + 1)[0] for l in idx])
    np.testing.assert_array_equal(got, T.OracleMesh(V, F).signed_distance(pos))
    # one shard of an 8-rank decomposition agrees with the same nodes of the full run
    cnt, stride = dg.shard_layout(g, 3, 8)
    packed = torch.empty(stride, dtype=torch.float64, device="cuda")
    m.sample_shard_device(g, 3, 8, packed.data_ptr(), stream=s)
    gathered = torch.zeros(8 * stride, dtype=torch.float64, device="cuda")
    gathered[3 * stride:4 * stride] = packed
    field = torch.full((n,), float("nan"), dtype=torch.float64, device="cuda")
    dg.unpack_shards_device(g, 8, gathered.data_ptr(), stride, field.data_ptr(), stream=s)
    torch.cuda.synchronize()
    mine = field != 0.0          # nodes of the other ranks were gathered as 0
    assert abs(int(mine.sum().item()) - cnt) <= 64  # (a handful of rank-3 nodes may be exactly 0)
    assert torch.equal(field[mine], out[mine])


def test_sliver_band_around_the_filter_threshold(dg, monkeypatch):
    """A triangle soup of slivers whose shape ratio area2 / lmax^2 is swept over 1e-6 .. 1e-2 -- both sides of the
    float filter's 1e-4 threshold (dg_geom.h: below it a triangle is degenerate for the filter and its wave runs the
    exact traversal) --, acute and obtuse, with points 1e-6 .. 1 side lengths off the planes, over the sharp corners
    and beyond them: the filtered kernel, the exact kernel and the oracle must agree bit for bit, through the point
    entry (K1p, binned and not) and on a lattice (K1).  (The k1_variant fixture runs this with either kernel forced;
    the test itself forces both once more so that one run compares them directly.)"""
    import sys
    import os
    sys.path.insert(0, os.path.join(T.ROOT, "tests", "perf"))
    import filter_campaign as fc
    rng = np.random.default_rng(99)
    tri = np.concatenate([fc.sliver_band(rng, 3000, lo, hi, 1.0) for lo, hi in ((1e-6, 1e-4), (1e-4, 3e-4), (3e-4, 1e-2))])
    rho, _ = fc.shape_ratio(tri)
    assert (rho < 1e-4).sum() > 2000 and ((rho > 1e-4) & (rho < 1e-3)).sum() > 2000
    V = tri.reshape(-1, 3)
    F = np.arange(len(V), dtype=np.uint32).reshape(-1, 3)
    P = np.concatenate([fc.adversarial_points(rng, tri, 60000), rng.uniform(-1.5, 1.5, size=(20000, 3))])
    want = np.abs(T.OracleMesh(V, F).signed_distance(P))
    got = {}
    for fast in ("0", "1"):
        T.force(monkeypatch, k1_fast=fast)
        m = dg.Mesh(V, F)
        for binning in ("1", "0"):
            T.force(monkeypatch, k1p_binning=binning)
            d = np.abs(m.signed_distance(P))     # (a soup has no inside: only the magnitude is defined by the reference)
            np.testing.assert_array_equal(d, want)
        dom = np.array([-1.2, -1.2, -1.2, 1.2, 1.2, 1.2])
        got[fast] = m.sample_nodes(dg.grid_desc(dom[:3], dom[3:], [24, 20, 28]))
    np.testing.assert_array_equal(got["0"], got["1"])
    np.testing.assert_array_equal(np.abs(got["1"]), np.abs(T.OracleMesh(V, F).sample_nodes(np.array([-1.2, -1.2, -1.2, 1.2, 1.2, 1.2]), [24, 20, 28])))
