"""GPU parity tests: the HIP kernels, called through the C ABI (libdiscregrid_hip.so), against
the CPU oracle and the committed golden vectors of the unmodified reference.

Tolerance (BASELINE.json north_star): coefficients and interpolated values within 1e-10
RELATIVE of the reference in double precision.  The kernels keep the reference's operation
order, so the tests additionally assert BIT equality wherever no exact tie between triangles
is involved; the relative bound is what is contractually required."""
import os
import time

import numpy as np
import pytest

import dgtest as T

from conftest import k1_variant_fixture

pytestmark = pytest.mark.gpu
DBL_MAX = np.finfo(np.float64).max
TOL = 1e-10
k1_variant = k1_variant_fixture()
MESHES = {"box": T.box_mesh, "ico8": lambda: T.icosphere(8), "torus": T.torus, "bunny": T.bunny_mesh}


@pytest.fixture(scope="module")
def dg():
    import discregrid_amd
    discregrid_amd.load_library()          # raises if the HIP extension is missing
    assert discregrid_amd.device_count() >= 1, "no HIP device: the product has no CPU path"
    return discregrid_amd


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def grid_of(dg, dom, res):
    return dg.grid_desc(dom[:3], dom[3:], res)


def assert_parity(got, want, what):
    rel = T.rel_err(got, want)
    nbad = int((got != want).sum())
    print("%s: max rel err %.3e, %d / %d not bit-equal" % (what, rel, nbad, len(want)))
    assert rel <= TOL, "%s: relative error %.3e > %g" % (what, rel, TOL)
    np.testing.assert_array_equal(got == DBL_MAX, want == DBL_MAX)
    return nbad


def test_box_cdf_known_answer(dg, tmp_path):
    """The reference's only golden file, reproduced byte for byte from GPU coefficients."""
    V, F = T.box_mesh()
    dom = T.oracle_default_domain(V)
    coeffs = dg.Mesh(V, F).sample_nodes(grid_of(dg, dom, [5, 5, 5]))
    p = str(tmp_path / "box_gpu.cdf")
    assert T.oracle_write_cdf(p, dom, [5, 5, 5], [coeffs]) == 27040
    assert open(p, "rb").read() == open(os.path.join(T.GOLDEN, "box.cdf"), "rb").read()


@pytest.mark.parametrize("name", list(MESHES))
def test_sampling_vs_golden(dg, golden, name):
    V, F = MESHES[name]()
    dom, res = golden[name + "_domain"], golden[name + "_res"]
    got = dg.Mesh(V, F).sample_nodes(grid_of(dg, dom, res))
    assert assert_parity(got, golden[name + "_coeffs"], name) == 0   # bit-exact


@pytest.mark.parametrize("name", ["ico8", "torus", "bunny"])
def test_ranges_masks_invert(dg, name):
    V, F = MESHES[name]()
    dom = T.oracle_default_domain(V)
    res = [9, 6, 11]
    g = grid_of(dg, dom, res)
    n = T.n_nodes(res)
    ref = T.OracleMesh(V, F).sample_nodes(dom, res)
    m = dg.Mesh(V, F)
    rng = np.random.default_rng(11)
    cuts = sorted(rng.integers(0, n, size=6).tolist() + [0, n])
    for b, e in zip(cuts[:-1], cuts[1:]):
        np.testing.assert_array_equal(m.sample_nodes(g, b, e), ref[b:e])
    assert len(m.sample_nodes(g, 7, 7)) == 0
    mask = rng.integers(0, 2, size=n).astype(np.uint8)
    got = m.sample_nodes(g, mask=mask)
    np.testing.assert_array_equal(got[mask == 1], ref[mask == 1])
    assert (got[mask == 0] == DBL_MAX).all()
    np.testing.assert_array_equal(m.sample_nodes(g, invert=True), -1.0 * ref)
    with pytest.raises(dg.DiscregridError):
        m.sample_nodes(g, 0, n + 1)


@pytest.mark.parametrize("name", list(MESHES))
def test_heavy_brick_split_is_bit_exact(dg, golden, monkeypatch, name):
    """Bricks that exhaust their work budget are parked and finished by k_heavy_subtrees /
    k_heavy_finish.  Forced here with a tiny budget (DG_FORCE=heavy_work=...) so that (almost) every brick
    takes that path, with only 3 slots (all other heavy bricks carry on unsplit), and switched
    off: always the bits of the golden vectors; masks, inversion and shards included."""
    V, F = MESHES[name]()
    dom, res = golden[name + "_domain"], golden[name + "_res"]
    want = golden[name + "_coeffs"]
    g = grid_of(dg, dom, res)
    m = dg.Mesh(V, F)
    for slots, work in (("256", "4"), ("3", "1"), ("256", "60"), ("0", "4")):
        T.force(monkeypatch, heavy_slots=slots, heavy_work=work)
        for _ in range(2):   # the second launch reuses the scratch of the first
            assert assert_parity(m.sample_nodes(g), want, name) == 0
        heavy, split = m.last_heavy_bricks()
        if slots == "0":
            assert (heavy, split) == (0, 0)
        elif work != "60":
            assert heavy >= split == min(int(slots), heavy) > 0, (heavy, split)   # the path really ran
    T.force(monkeypatch, heavy_slots=256, heavy_work=4)
    rng = np.random.default_rng(5)
    mask = rng.integers(0, 2, size=len(want)).astype(np.uint8)
    got = m.sample_nodes(g, mask=mask, invert=True)
    np.testing.assert_array_equal(got[mask == 1], -1.0 * want[mask == 1])
    assert (got[mask == 0] == DBL_MAX).all()
    b, e = len(want) // 3, 2 * len(want) // 3 + 5
    np.testing.assert_array_equal(m.sample_nodes(g, b, e), want[b:e])


def test_signed_distance_points(dg, golden):
    for name in MESHES:
        V, F = MESHES[name]()
        P = golden[name + "_P"]
        d, tri, ent, near = dg.Mesh(V, F).signed_distance(P, full=True)
        assert_parity(d, golden[name + "_sd"], name + " signed_distance")
        np.testing.assert_array_equal(d, golden[name + "_sd"])
        same = tri == golden[name + "_tri"]
        np.testing.assert_array_equal(ent[same], golden[name + "_ent"][same])
        np.testing.assert_array_equal(near[same], golden[name + "_near"][same])
        # where the triangle differs from the reference's, the two are tied under the reference's own
        # acceptance rule (same stored distance, checked with the reference's per-triangle arithmetic)
        T.assert_exact_ties(V, F, P, tri, golden[name + "_tri"])
        # the single-point host evaluator of the ABI returns the same bits as the batch
        m = dg.Mesh(V, F)
        for i in range(0, len(P), 37):
            d1, t1, e1, n1 = m.signed_distance_point(P[i], full=True)
            assert d1 == d[i]
            if t1 == tri[i]:
                assert e1 == ent[i] and np.array_equal(n1, near[i])
        T.assert_exact_ties(V, F, P[::37], [m.signed_distance_point(p, full=True)[1] for p in P[::37]], tri[::37])


def test_signed_distance_binned_launch(dg, monkeypatch):
    """K1p groups unordered points into compact tiles before the packet traversal (device-side
    decision).  Distances are order-independent, so the binned launch must return the bits of the
    plain launch for random, line-ordered, far-away and duplicated inputs; the winning triangle
    may differ only where several triangles tie exactly."""
    V, F = T.bunny_mesh()
    m = dg.Mesh(V, F)
    lo, hi = V.min(axis=0), V.max(axis=0)
    rng = np.random.default_rng(21)
    P = rng.uniform(lo - 0.3 * (hi - lo), hi + 0.3 * (hi - lo), size=(50021, 3))
    P[:3000] = rng.uniform(lo - 40 * (hi - lo), hi + 40 * (hi - lo), size=(3000, 3))     # far outside the tile grid
    P[3000:4000] = P[3000]                                                              # duplicates
    line = np.linspace(lo - 0.1, hi + 0.1, 20000)                                       # one long line
    for Q in (P, line, P[np.argsort(P[:, 0])]):
        T.force(monkeypatch, k1p_binning=0)
        d0, t0, e0, n0 = m.signed_distance(Q, full=True)
        T.force(monkeypatch, k1p_binning=1)
        d1, t1, e1, n1 = m.signed_distance(Q, full=True)
        np.testing.assert_array_equal(d1, d0)
        # exact ties (nearest point on a shared edge or vertex: identical d^2 from several triangles)
        # are broken by visiting order, which depends on which points share a wave: the triangle may
        # differ, the nearest point is the same point
        same = t1 == t0
        T.assert_exact_ties(V, F, Q, t1, t0)      # proven with the reference's own per-triangle arithmetic
        np.testing.assert_array_equal(e1[same], e0[same])
        np.testing.assert_array_equal(n1[same], n0[same])
        assert np.abs(n1 - n0).max() <= 1e-12 * np.abs(hi - lo).max()
    np.testing.assert_array_equal(d1[:2000], T.OracleMesh(V, F).signed_distance(P[np.argsort(P[:, 0])][:2000]))


def test_far_and_on_surface_queries(dg):
    V, F = T.icosphere(6)
    for shift, scale in ((0.0, 1.0), (1000.0, 1.0), (-3.0e4, 250.0), (0.5, 1e-3)):
        W = V * scale + shift
        om, m = T.OracleMesh(W, F), dg.Mesh(W, F)
        rng = np.random.default_rng(4)
        P = np.concatenate([
            rng.uniform(-2, 2, size=(400, 3)) * scale + shift,
            rng.uniform(-1, 1, size=(50, 3)) * scale * 1e4 + shift,
            W[:100], 0.5 * (W[F[:50, 0]] + W[F[:50, 1]]), (W[F[:50, 0]] + W[F[:50, 1]] + W[F[:50, 2]]) / 3.0])
        a, b = m.signed_distance(P), om.signed_distance(P)
        np.testing.assert_array_equal(np.abs(a), np.abs(b))
        off = np.abs(b) > 1e-7 * (scale + abs(shift))  # sqrt(eps): d^2 cancels to ~1e-16 there
        np.testing.assert_array_equal(a[off], b[off])


@pytest.mark.parametrize("name", list(MESHES))
def test_interpolate_vs_golden(dg, golden, name, monkeypatch):
    dom, res = golden[name + "_domain"], golden[name + "_res"]
    coeffs, P = golden[name + "_coeffs"], golden[name + "_P"]
    g = grid_of(dg, dom, res)
    f = dg.Field(g, coeffs)
    phi, grad = f.interpolate(P, grad=True)
    assert assert_parity(phi, golden[name + "_phi"], name + " interpolate") == 0
    inside = golden[name + "_phi"] != DBL_MAX
    assert T.rel_err(grad[inside], golden[name + "_grad"][inside]) <= TOL
    np.testing.assert_array_equal(grad[inside], golden[name + "_grad"][inside])
    assert (grad[~inside] == 0).all()
    np.testing.assert_array_equal(f.interpolate(P), golden[name + "_phi"])
    # cell-major device copy: bit-identical results -- through the cooperative row kernel (default: no binning, any
    # query order, batch sizes that are no multiples of a wave) and through the per-lane kernels (DG_FORCE=k2_rows=0)
    f.build_cell_major()
    for rows in ("1", "0"):
        T.force(monkeypatch, k2_rows=rows)
        phi2, grad2 = f.interpolate(P, grad=True)
        np.testing.assert_array_equal(phi2, phi)
        np.testing.assert_array_equal(grad2, grad)
        np.testing.assert_array_equal(f.interpolate(P), golden[name + "_phi"])
        for m in (1, 63, 65, 1000):
            np.testing.assert_array_equal(f.interpolate(P[:m]), golden[name + "_phi"][:m])
    T.force(monkeypatch, k2_rows=None)
    f.drop_cell_major()
    np.testing.assert_array_equal(f.interpolate(P), golden[name + "_phi"])
    # tile-major device copy (4^3-cell tiles of 736 doubles): bit-identical results, takes precedence over cell-major
    f.build_tile_major()
    phi3, grad3 = f.interpolate(P, grad=True)
    np.testing.assert_array_equal(phi3, phi)
    np.testing.assert_array_equal(grad3, grad)
    f.build_cell_major()
    np.testing.assert_array_equal(f.interpolate(P), golden[name + "_phi"])
    f.drop_tile_major()
    f.drop_cell_major()
    np.testing.assert_array_equal(f.interpolate(P), golden[name + "_phi"])
    # band-limited cell-major copy (round 4): rows for the cells that reach into a value band, the plain gather for the others
    # in the same launch -- wide, narrow and empty bands, any batch size, dropped again
    fin = coeffs[coeffs != DBL_MAX]
    for lo, hi in ((-0.05, 0.05), (float(np.median(fin)), float(fin.max())), (float(fin.max()) + 1.0, float(fin.max()) + 2.0),
                   (float(fin.min()) - 1.0, float(fin.max()) + 1.0)):
        rows = f.build_cell_major_band(lo, hi)
        assert f.info()["band_rows"] == rows and 0 <= rows <= int(np.prod(res))
        phi4, grad4 = f.interpolate(P, grad=True)
        np.testing.assert_array_equal(phi4, phi)
        np.testing.assert_array_equal(grad4, grad)
        for m in (1, 63, 65, 1000):
            np.testing.assert_array_equal(f.interpolate(P[:m]), golden[name + "_phi"][:m])
        f.drop_cell_major()
        assert f.info()["band_rows"] == 0
    assert rows == int(np.prod(res))            # (the last band holds every value: every cell has a row)
    # table mode, removed cells, DBL_MAX coefficients
    cells = T.oracle_cell_table(res)
    cmap = np.arange(len(cells), dtype=np.uint32)
    np.testing.assert_array_equal(dg.Field(g, coeffs, cells, cmap).interpolate(P), golden[name + "_phi"])
    cmap2 = cmap.copy()
    cmap2[::3] = 0xFFFFFFFF
    c2 = coeffs.copy()
    c2[::7] = DBL_MAX
    f2 = dg.Field(g, c2, cells, cmap2)
    a, ga = f2.interpolate(P, grad=True)
    b, gb = T.oracle_interpolate(dom, res, c2, P, grad=True, cells=cells, cell_map=cmap2)
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(ga[b != DBL_MAX], gb[b != DBL_MAX])
    f2.build_cell_major()
    a2, ga2 = f2.interpolate(P, grad=True)
    np.testing.assert_array_equal(a2, a)
    np.testing.assert_array_equal(ga2, ga)
    f2.drop_cell_major()
    f2.build_cell_major_band(-0.1, 0.1)      # the band copy of a table-mode field with removed cells and "no value" coefficients
    a3, ga3 = f2.interpolate(P, grad=True)
    np.testing.assert_array_equal(a3, a)
    np.testing.assert_array_equal(ga3, ga)
    with pytest.raises(dg.DiscregridError):
        f2.build_tile_major()   # unreduced fields only


@pytest.mark.parametrize("n_meshes", [1, 2, 3])
def test_multi_device_host_path(dg, golden, monkeypatch, n_meshes):
    """dg_sdf_sample_nodes_multi: chunks of the lattice dealt round-robin to several mesh handles, each
    with its own host thread and copy pipeline writing straight into the caller's array.  (One GPU
    here: the handles share the device, which exercises the same code.)  Small chunks force many
    pipeline turns; masks, inversion and sub-ranges included."""
    V, F = T.torus()
    dom, res = golden["torus_domain"], golden["torus_res"]
    want = golden["torus_coeffs"]
    g = grid_of(dg, dom, res)
    meshes = [dg.Mesh(V, F) for _ in range(n_meshes)]
    for chunk, direct in (("2000", "2"), ("2000", "0"), ("100000000", "2")):
        T.force(monkeypatch, host_chunk_nodes=chunk, host_direct=direct)   # 2: results DMA'd straight into the array, 0: staged
        np.testing.assert_array_equal(dg.sample_nodes_multi(meshes, g), want)
        rng = np.random.default_rng(8)
        mask = rng.integers(0, 2, size=len(want)).astype(np.uint8)
        got = dg.sample_nodes_multi(meshes, g, mask=mask, invert=True)
        np.testing.assert_array_equal(got[mask == 1], -1.0 * want[mask == 1])
        assert (got[mask == 0] == DBL_MAX).all()
        b, e = 1234, len(want) - 777
        np.testing.assert_array_equal(dg.sample_nodes_multi(meshes, g, b, e, mask=mask[b:e])[mask[b:e] == 1],
                                      want[b:e][mask[b:e] == 1])
        # the single-mesh host path shares the pipeline code
        np.testing.assert_array_equal(meshes[0].sample_nodes(g, mask=mask)[mask == 1], want[mask == 1])
    with pytest.raises(dg.DiscregridError):
        dg.sample_nodes_multi(meshes, g, 0, len(want) + 1)


@pytest.mark.parametrize("name", ["torus", "bunny"])
def test_interpolate_binned_path(dg, golden, monkeypatch, name):
    """K2 with query binning (tile-by-tile processing of unordered queries; the ordered/unordered
    decision is taken on the device): forced on for a small batch, results and gradients must be
    the bits of the direct path for random, cell-sorted, out-of-domain and duplicated queries."""
    dom, res = golden[name + "_domain"], golden[name + "_res"]
    coeffs = golden[name + "_coeffs"]
    f = dg.Field(grid_of(dg, dom, res), coeffs)
    rng = np.random.default_rng(3)
    lo, hi = dom[:3], dom[3:]
    P = rng.uniform(lo - 0.05 * (hi - lo), hi + 0.05 * (hi - lo), size=(70001, 3))   # some outside
    P[1000:2000] = P[1000]                                                            # duplicates
    cell = np.floor((np.clip(P, lo, hi) - lo) / (hi - lo) * res).astype(np.int64)
    order = np.lexsort((cell[:, 0], cell[:, 1], cell[:, 2]))
    T.force(monkeypatch, k2_binning=0)
    want = {k: f.interpolate(Q, grad=True) for k, Q in (("random", P), ("sorted", P[order]))}
    np.testing.assert_array_equal(want["random"][0], T.oracle_interpolate(dom, res, coeffs, P))
    T.force(monkeypatch, k2_binning=2)
    for k, Q in (("random", P), ("sorted", P[order]), ("random", P)):
        phi, grad = f.interpolate(Q, grad=True)
        np.testing.assert_array_equal(phi, want[k][0])
        np.testing.assert_array_equal(grad, want[k][1])
        np.testing.assert_array_equal(f.interpolate(Q), want[k][0])
    np.testing.assert_array_equal(f.interpolate(P[:5]), want["random"][0][:5])            # tiny batch


@pytest.mark.parametrize("res", [[9, 14, 6], [16, 8, 24], [13, 21, 7], [1, 1, 1], [33, 2, 17]])
def test_interpolate_staged_tiles_path(dg, monkeypatch, res):
    """K2 on the plain layout, round 6: counting sort by tile of 8^3 cells + the gather that serves a tile's queries from an LDS image
    of the tile (k_interpolate_tiles).  Lattices whose resolution is no multiple of the tile (clipped tiles), smaller than one tile,
    with "no value" coefficients; queries inside, outside, ON cell / tile / domain faces, NaN, duplicated, and 5000 of them in ONE
    cell (several work items for one tile); with and without the XCD-aware item order, the radix-sorted per-lane gather of rounds 1-5, the default routing: all the bits of the oracle's interpolate, value and gradient."""
    rng = np.random.default_rng(sum(res))
    lo, hi = np.array([-0.7, 0.1, -1.3]), np.array([1.1, 2.3, 0.9])
    dom = np.concatenate([lo, hi])
    n = T.n_nodes(res)
    coeffs = rng.normal(size=n)
    coeffs[rng.integers(0, n, size=max(1, n // 200))] = np.finfo(np.float64).max
    P = rng.uniform(lo - 0.03 * (hi - lo), hi + 0.03 * (hi - lo), size=(60001, 3))
    cells = (hi - lo) / np.array(res)
    faces = lo + cells * rng.integers(0, np.array(res) + 1, size=(4000, 3))      # exactly on lattice planes (cell and tile faces, domain faces)
    P[2000:6000] = np.where(rng.random((4000, 3)) < 0.6, faces, P[2000:6000])
    P[6000:11000] = lo + cells * (np.minimum(np.array(res) - 1, [3, 5, 2]) + rng.random((5000, 3)))   # one cell, many queries
    P[11000:11500] = P[11000]
    P[11500:11510] = np.nan
    P[11510] = hi
    P[11511] = lo
    want = T.oracle_interpolate(dom, res, coeffs, P, grad=True)
    inside = np.all((P >= lo) & (P <= hi), axis=1)        # (the reference leaves the gradient of a query outside the domain untouched;
    want[1][~inside] = 0.0                                #  the batched evaluator writes zeros)
    assert (want[0][~inside] == np.finfo(np.float64).max).all() and 0.7 < inside.mean() < 0.95
    f = dg.Field(grid_of(dg, dom, res), coeffs)
    for force in (dict(k2_tiles=2), dict(k2_tiles=2, k2_tile_chunk=0), dict(k2_tiles=2, k2_tile_chunk=3), dict(k2_tiles=0), dict(k2_tiles=1)):
        T.force(monkeypatch, k2_binning=2, **force)
        for _ in range(2):        # (the second call runs on the prediction of the first)
            phi, grad = f.interpolate(P, grad=True)
            np.testing.assert_array_equal(phi, want[0], err_msg=str(force))
            np.testing.assert_array_equal(grad, want[1], err_msg=str(force))
            np.testing.assert_array_equal(f.interpolate(P), want[0], err_msg=str(force))
        T.force(monkeypatch, **{k: None for k in force})
    # ORDERED batches through the same paths -- queries in z-order of their cells (as an SPH code's spatially sorted particles arrive), in row
    # order of their cells, and half ordered, half not: the device-side "is this batch ordered?" verdict of one call routes the next
    cell = np.clip(np.floor((np.clip(np.nan_to_num(P), lo, hi) - lo) / cells), 0, np.array(res) - 1).astype(np.int64)   # (the NaN queries: anywhere)
    zkey = np.zeros(len(P), dtype=np.int64)
    for b in range(10):
        for d in range(3):
            zkey |= ((cell[:, d] >> b) & 1) << (3 * b + d)
    rowkey = (cell[:, 2] * res[1] + cell[:, 1]) * res[0] + cell[:, 0]
    orders = {"z-order": np.argsort(zkey, kind="stable"), "rows": np.argsort(rowkey, kind="stable"), "random": np.arange(len(P))}
    orders["half"] = np.concatenate([orders["z-order"][: len(P) // 2], orders["random"][len(P) // 2:]])
    T.force(monkeypatch, k2_binning=2, k2_tiles=2)
    for name, o in orders.items():
        Q = np.ascontiguousarray(P[o])
        phi, grad = f.interpolate(Q, grad=True)
        np.testing.assert_array_equal(phi, want[0][o], err_msg=name)
        np.testing.assert_array_equal(grad, want[1][o], err_msg=name)
        np.testing.assert_array_equal(f.interpolate(Q), want[0][o], err_msg=name)
        for m in (1, 255, 257, 2049):
            np.testing.assert_array_equal(f.interpolate(Q[:m]), want[0][o][:m], err_msg=name)
    T.force(monkeypatch, k2_binning=None, k2_tiles=None)
    f.close()


@pytest.mark.parametrize("nranks", [2, 3, 8, 32])
def test_shards_on_one_gpu_equal_unsharded(dg, torch, nranks, monkeypatch):
    """Multi-GPU path exercised on one device: every rank's shard is computed in turn, the
    all-gather is a concatenation, the unpack kernel restores reference order -- bit for bit
    the unsharded result."""
    V, F = T.torus()
    dom = T.oracle_default_domain(V)
    res = [21, 18, 26]
    g = grid_of(dg, dom, res)
    m = dg.Mesh(V, F)
    n = dg.n_nodes(g)
    ref = m.sample_nodes(g)
    if nranks == 3:
        T.force(monkeypatch, heavy_work=8)   # shards through the heavy-brick path as well
    stride = dg.shard_layout(g, 0, nranks)[1]
    gathered = torch.full((nranks * stride,), float("nan"), dtype=torch.float64, device="cuda")
    total = 0
    for r in range(nranks):
        cnt, st = dg.shard_layout(g, r, nranks)
        assert st == stride
        total += cnt
        m.sample_shard_device(g, r, nranks, gathered.data_ptr() + 8 * r * stride,
                              stream=torch.cuda.current_stream().cuda_stream)
    assert total == n
    field = torch.empty(n, dtype=torch.float64, device="cuda")
    dg.unpack_shards_device(g, nranks, gathered.data_ptr(), stride, field.data_ptr(),
                            stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(field.cpu().numpy(), ref)
    np.testing.assert_array_equal(ref, T.OracleMesh(V, F).sample_nodes(dom, res))
    # the same buffer unpacked slot range by slot range (pieced gather)
    field.fill_(float("nan"))
    cuts = sorted({0, nranks // 3, nranks // 2, nranks})
    for r0, r1 in zip(cuts[:-1], cuts[1:]):
        dg.unpack_shard_range_device(g, nranks, gathered.data_ptr(), stride, r0, r1, field.data_ptr(),
                                     stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(field.cpu().numpy(), ref)
    with pytest.raises(dg.DiscregridError):
        dg.unpack_shard_range_device(g, nranks, gathered.data_ptr(), stride, 1, nranks + 1, field.data_ptr())


def test_config2_bunny_128(dg, torch, golden):
    """BASELINE config 2: Stanford bunny (69 630 triangles), 128^3 grid = 14 926 977 nodes."""
    V, F = T.bunny_mesh()
    dom = golden["bunny128_domain"]
    res = [128] * 3
    g = grid_of(dg, dom, res)
    m = dg.Mesh(V, F)
    n = dg.n_nodes(g)
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    t0 = time.time()
    m.sample_nodes_device(g, 0, n, out.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print("bunny 128^3: %.1f ms, %.1f Mnodes/s" % (dt * 1e3, n / dt / 1e6))
    got = out.cpu().numpy()
    idx = golden["bunny128_lattice_idx"]
    np.testing.assert_array_equal(got[idx], golden["bunny128_lattice_sd"])   # reference itself
    sel = np.random.default_rng(8).integers(0, n, size=20000)
    om = T.OracleMesh(V, F)
    pos = T.oracle_node_positions(dom, res)[sel]
    want = om.signed_distance(pos)
    assert_parity(got[sel], want, "bunny128 random nodes")
    np.testing.assert_array_equal(got[sel], want)


def test_config3_icosphere_256(dg, torch, golden):
    """BASELINE config 3 (headline): icosphere nu=71 (100 820 triangles), 256^3 grid =
    118 425 857 nodes, full size.  Checked against (a) the reference's values on a strided
    lattice sample, (b) the analytic sphere distance everywhere, (c) bit equality of the
    sharded and the unsharded evaluation."""
    V, F = T.icosphere(71)
    dom = golden["ico71_domain"]
    res = [256] * 3
    g = grid_of(dg, dom, res)
    m = dg.Mesh(V, F)
    n = dg.n_nodes(g)
    assert n == 118425857
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    m.sample_nodes_device(g, 0, n, out.data_ptr(), stream=s)       # warm-up
    torch.cuda.synchronize()
    t0 = time.time()
    m.sample_nodes_device(g, 0, n, out.data_ptr(), stream=s)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print("icosphere 256^3: %.1f ms, %.1f Mnodes/s" % (dt * 1e3, n / dt / 1e6))
    idx = torch.from_numpy(golden["ico71_lattice_idx"].astype(np.int64)).cuda()
    np.testing.assert_array_equal(out[idx].cpu().numpy(), golden["ico71_lattice_sd"])
    # (b) analytic: |sdf - (|x| - 1)| <= sagitta of a facet (edge^2 / 8 ~ 3e-5)
    got = out.cpu().numpy()
    for b, e in ((0, 2000000), (n // 2, n // 2 + 2000000), (n - 2000000, n)):
        pos = T.oracle_node_positions(dom, res, b, e)
        assert np.abs(got[b:e] - (np.linalg.norm(pos, axis=1) - 1.0)).max() < 1e-4
    # (c) sharded == unsharded
    nr = 4
    stride = dg.shard_layout(g, 0, nr)[1]
    gathered = torch.empty(nr * stride, dtype=torch.float64, device="cuda")
    for r in range(nr):
        m.sample_shard_device(g, r, nr, gathered.data_ptr() + 8 * r * stride, stream=s)
    field = torch.empty(n, dtype=torch.float64, device="cuda")
    dg.unpack_shards_device(g, nr, gathered.data_ptr(), stride, field.data_ptr(), stream=s)
    torch.cuda.synchronize()
    assert torch.equal(field, out)
    # random nodes against the oracle
    sel = np.random.default_rng(9).integers(0, n, size=3000)
    pos = np.stack([T.oracle_node_positions(dom, res, int(l), int(l) + 1)[0] for l in sel])
    want = T.OracleMesh(V, F).signed_distance(pos)
    assert_parity(got[sel], want, "ico71 256^3 random nodes")
    np.testing.assert_array_equal(got[sel], want)


def test_k2_builds_the_cell_major_copy_of_an_owned_field_by_itself(dg, monkeypatch):
    """A field created from host coefficients (the library owns its device copy: it cannot change) that receives a
    large batch gets its cell-major copy built by dg_interpolate_batch itself, once, and from then on every batch --
    whatever the order of its queries -- runs through the cooperative row kernel: same bits as the plain layout and
    as the oracle; the memory shows up and, after dg_field_drop_cell_major, stays away."""
    import torch
    V, F = T.torus(40, 20)
    dom = T.oracle_default_domain(V)
    res = [96, 96, 96]
    g = grid_of(dg, dom, res)
    coeffs = dg.Mesh(V, F).sample_nodes(g)
    assert coeffs.nbytes >= (32 << 20)
    P = T.uniform_points(99, 300_000, dom[:3] - 0.05, dom[3:] + 0.05)   # unordered, a few outside the domain
    monkeypatch.setenv("DG_K2_AUTO_CELL_MAJOR_MB", "0")
    f0 = dg.Field(g, coeffs)
    phi0, grad0 = f0.interpolate(P, grad=True)
    monkeypatch.delenv("DG_K2_AUTO_CELL_MAJOR_MB")
    f1 = dg.Field(g, coeffs)
    torch.cuda.synchronize()
    free_before = torch.cuda.mem_get_info()[0]
    phi1, grad1 = f1.interpolate(P, grad=True)
    torch.cuda.synchronize()
    free_after = torch.cuda.mem_get_info()[0]
    np.testing.assert_array_equal(phi1, phi0)
    np.testing.assert_array_equal(grad1, grad0)
    copy_bytes = 256 * int(np.prod(res))
    assert free_before - free_after >= 0.9 * copy_bytes          # the copy was built ...
    np.testing.assert_array_equal(f1.interpolate(P[::-1].copy()), phi0[::-1])
    assert free_after - torch.cuda.mem_get_info()[0] < 0.5 * copy_bytes   # ... once
    want = T.oracle_interpolate(dom, res, coeffs, P[:5000])
    np.testing.assert_array_equal(phi1[:5000], want)
    f1.drop_cell_major()
    torch.cuda.synchronize()
    free_dropped = torch.cuda.mem_get_info()[0]
    assert free_dropped - free_after >= 0.9 * copy_bytes
    np.testing.assert_array_equal(f1.interpolate(P), phi0)
    torch.cuda.synchronize()
    assert free_dropped - torch.cuda.mem_get_info()[0] < 0.5 * copy_bytes  # the caller dropped it: it stays away
