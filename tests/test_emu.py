"""CPU lock-step emulation of the HIP kernels (tests/emu/wave_emu.cpp) against the oracle.
Validates, without a GPU, everything about the product that is not the GPU's own
instruction execution: the host-built BVH / triangle packets / pseudonormals, the
conservative float box test, the packet traversal's pruning logic, the per-lane arithmetic
in dg_geom.h, the brick decomposition (every node exactly once, incl. the XCD block remap),
flat ranges, predicate masks, shard packing and unpacking, and the K2 per-query body."""
import numpy as np
import pytest

import dgtest as T
import emu

DBL_MAX = np.finfo(np.float64).max
MESHES = {"box": T.box_mesh, "ico8": lambda: T.icosphere(8), "torus": T.torus, "bunny": T.bunny_mesh}


@pytest.mark.parametrize("name", list(MESHES))
def test_bvh_structure_and_pseudonormals(name):
    V, F = MESHES[name]()
    for leaf in (1, 4, 8, 16):
        em = emu.EmuMesh(V, F, max_leaf=leaf)
        assert em.check() == 0
    em = emu.EmuMesh(V, F)
    c = T.OracleMesh(V, F).construction()
    pn = em.pseudonormals()
    np.testing.assert_array_equal(pn[:, 6], c["pn_tri"])
    np.testing.assert_array_equal(pn[:, 3:6], c["pn_edge"])
    for k in range(3):
        np.testing.assert_array_equal(pn[:, k], c["pn_vert"][F[:, k]])
    assert em.info()["flags"] == 0


def test_pseudonormals_of_an_open_non_manifold_mesh():
    """Edges with one and with three or four incident triangles (fins on an icosphere with holes), enough triangles for the
    threaded parts of the builder: the per-vertex and per-edge SUMS depend on the order of the additions -- triangle order, as
    the reference's single loop forms them (TriangleMeshDistance.h:336-428) -- and have to come out bit for bit."""
    V, F = T.icosphere(30)                       # 18 000 triangles
    rng = np.random.default_rng(3)
    keep = np.ones(len(F), dtype=bool)
    keep[rng.choice(len(F), 400, replace=False)] = False          # holes: edges with one triangle
    fins = []
    Vx = [V]
    for t in rng.choice(np.nonzero(keep)[0], 600, replace=False):  # fins: a third (and sometimes fourth) triangle on an edge
        a, b = F[t, 0], F[t, 1]
        for _ in range(int(rng.integers(1, 3))):
            apex = 1.2 * 0.5 * (V[a] + V[b]) + 0.05 * rng.normal(size=3)
            Vx.append(apex[None, :])
            fins.append([a, b, sum(len(v) for v in Vx) - 1])
    V2 = np.concatenate(Vx)
    F2 = np.concatenate([F[keep], np.asarray(fins, dtype=F.dtype)])
    F2 = F2[rng.permutation(len(F2))]             # incident triangles far apart in triangle order
    em = emu.EmuMesh(V2, F2)
    c = T.OracleMesh(V2, F2).construction()
    pn = em.pseudonormals()
    np.testing.assert_array_equal(pn[:, 6], c["pn_tri"])
    np.testing.assert_array_equal(pn[:, 3:6], c["pn_edge"])
    for k in range(3):
        np.testing.assert_array_equal(pn[:, k], c["pn_vert"][F2[:, k]])
    assert em.info()["flags"] == 3               # open edges AND edges with more than two triangles


@pytest.mark.parametrize("name", list(MESHES))
def test_subtree_cut_covers_the_tree_once(name):
    """The <= 256 subtree roots heavy bricks are split over reach every triangle exactly once."""
    V, F = MESHES[name]()
    for leaf in (1, 8, 16):
        em = emu.EmuMesh(V, F, max_leaf=leaf)
        assert 1 <= em.n_subtrees() <= 256
        assert em.subtree_triangles() == len(F)
    if len(F) >= 256 * 16:
        assert emu.EmuMesh(V, F).n_subtrees() == 256


@pytest.fixture
def heavy_settings():
    yield emu.set_heavy
    emu.set_heavy()  # back to the product's defaults


@pytest.mark.parametrize("name", ["box", "ico8", "torus", "bunny"])
def test_heavy_brick_split_is_bit_exact(golden, heavy_settings, name):
    """Bricks that exhaust their work budget are parked, searched subtree by subtree and merged
    (k_heavy_subtrees / k_heavy_finish): same bits as the single-wave traversal, with the budget
    forced so low that (almost) every brick takes the split path, with too few slots (the rest
    carries on unsplit) and with the split disabled."""
    V, F = MESHES[name]()
    dom, res = golden[name + "_domain"], golden[name + "_res"]
    want = golden[name + "_coeffs"]
    em = emu.EmuMesh(V, F)
    for slots, work, expect_heavy in ((256, 4, True), (3, 1, True), (256, 60, None), (0, 4, False)):
        heavy_settings(slots, work)
        got = em.sample_range(dom, res, stats=True)
        assert (em.written == 1).all()
        np.testing.assert_array_equal(got, want)
        if em.n_subtrees() < 2:
            assert em.stats["heavy_bricks"] == 0
        elif expect_heavy is True:
            assert em.stats["heavy_bricks"] == min(slots, em.stats["heavy_bricks"]) > 0
        elif expect_heavy is False:
            assert em.stats["heavy_bricks"] == 0
    # masks, inversion and shards go through the same path
    heavy_settings(256, 4)
    rng = np.random.default_rng(5)
    mask = rng.integers(0, 2, size=len(want)).astype(np.uint8)
    got = em.sample_range(dom, res, mask=mask, invert=True)
    np.testing.assert_array_equal(got[mask == 1], -1.0 * want[mask == 1])
    assert (got[mask == 0] == DBL_MAX).all()
    for r in range(3):
        emu_field = em.sample_shard(dom, res, r, 3)
        assert (em.written == 1).all() and np.isfinite(emu_field).all()


def test_open_mesh_is_flagged():
    V, F = T.box_mesh()
    assert emu.EmuMesh(V, F[:-1]).info()["flags"] & 1


@pytest.mark.parametrize("name", list(MESHES))
def test_sampling_bit_exact_vs_golden(golden, name):
    V, F = MESHES[name]()
    dom, res = golden[name + "_domain"], golden[name + "_res"]
    em = emu.EmuMesh(V, F)
    got = em.sample_range(dom, res)
    assert (em.written == 1).all()          # every node exactly once
    np.testing.assert_array_equal(got, golden[name + "_coeffs"])


@pytest.mark.parametrize("name", ["ico8", "torus", "bunny"])
def test_ranges_masks_invert(name):
    V, F = MESHES[name]()
    dom = T.oracle_default_domain(V)
    res = [9, 6, 11]
    n = T.n_nodes(res)
    ref = T.OracleMesh(V, F).sample_nodes(dom, res)
    em = emu.EmuMesh(V, F)
    rng = np.random.default_rng(11)
    cuts = sorted(rng.integers(0, n, size=6).tolist() + [0, n])
    for b, e in zip(cuts[:-1], cuts[1:]):   # ragged ranges crossing class boundaries
        if e > b:
            np.testing.assert_array_equal(em.sample_range(dom, res, b, e), ref[b:e])
            assert (em.written == 1).all()
    np.testing.assert_array_equal(em.sample_range(dom, res, 5, 5), np.empty(0))
    mask = rng.integers(0, 2, size=n).astype(np.uint8)
    got = em.sample_range(dom, res, mask=mask)
    np.testing.assert_array_equal(got[mask == 1], ref[mask == 1])
    assert (got[mask == 0] == DBL_MAX).all()
    np.testing.assert_array_equal(em.sample_range(dom, res, invert=True), -1.0 * ref)


@pytest.mark.parametrize("nranks", [1, 2, 3, 4, 8, 24])
def test_shard_pack_unpack(nranks):
    V, F = T.torus()
    dom = T.oracle_default_domain(V)
    res = [10, 7, 13]
    ref = T.OracleMesh(V, F).sample_nodes(dom, res)
    em = emu.EmuMesh(V, F)
    parts = []
    for r in range(nranks):
        p = em.sample_shard(dom, res, r, nranks)
        assert (em.written == 1).all() and len(p) == emu.shard_count(res, r, nranks)
        parts.append(p)
    assert sum(len(p) for p in parts) == len(ref)
    stride = (max(len(p) for p in parts) + 63) // 64 * 64
    G = np.full(stride * nranks, np.nan)
    for r, p in enumerate(parts):
        G[r * stride:r * stride + len(p)] = p
    np.testing.assert_array_equal(emu.unpack(res, nranks, G, stride), ref)
    # slot ranges unpacked one after the other (pieced gather) fill the field exactly once
    field = np.full(len(ref), np.nan)
    cuts = sorted({0, nranks // 3, nranks // 2, nranks})
    for r0, r1 in zip(cuts[:-1], cuts[1:]):
        before = np.isnan(field).sum()
        emu.unpack_ranks(res, nranks, G, stride, r0, r1, field)
        assert before - np.isnan(field).sum() == sum(len(p) for p in parts[r0:r1])
    np.testing.assert_array_equal(field, ref)


def test_signed_distance_points(golden):
    for name in MESHES:
        V, F = MESHES[name]()
        P = golden[name + "_P"]
        d, tri, ent, near = emu.EmuMesh(V, F).signed_distance(P, full=True)
        np.testing.assert_array_equal(d, golden[name + "_sd"])
        # exact ties (shared vertices/edges, or equidistant features) may resolve to another
        # triangle than the reference's walk: same distance, and where the
        # triangle agrees everything agrees
        same = tri == golden[name + "_tri"]
        assert same.mean() > 0.5
        np.testing.assert_array_equal(ent[same], golden[name + "_ent"][same])
        np.testing.assert_array_equal(near[same], golden[name + "_near"][same])
        np.testing.assert_allclose(np.linalg.norm(P - near, axis=1), np.abs(d), rtol=1e-9, atol=1e-9)  # d^2 formula cancels near the surface


def test_far_and_degenerate_queries():
    """Points far outside the mesh (float box test must stay conservative), exactly on
    vertices / edges / faces (d = 0), and a mesh far from the origin."""
    V, F = T.icosphere(6)
    for shift, scale in ((0.0, 1.0), (1000.0, 1.0), (-3.0e4, 250.0), (0.5, 1e-3)):
        W = V * scale + shift
        om, em = T.OracleMesh(W, F), emu.EmuMesh(W, F)
        rng = np.random.default_rng(4)
        P = np.concatenate([
            rng.uniform(-2, 2, size=(400, 3)) * scale + shift,
            rng.uniform(-1, 1, size=(50, 3)) * scale * 1e4 + shift,     # very far
            W[:100],                                                      # on vertices
            0.5 * (W[F[:50, 0]] + W[F[:50, 1]]),                          # on edges
            (W[F[:50, 0]] + W[F[:50, 1]] + W[F[:50, 2]]) / 3.0,           # on faces
        ])
        a, b = em.signed_distance(P), om.signed_distance(P)
        np.testing.assert_array_equal(np.abs(a), np.abs(b))
        # the sign of a point lying ON the surface (|d| ~ 1e-12) depends on which of the
        # tied triangles wins; everywhere else it must agree
        off = np.abs(b) > 1e-7 * (scale + abs(shift))  # sqrt(eps): d^2 cancels to ~1e-16 there
        np.testing.assert_array_equal(a[off], b[off])


def test_interpolate_body_bit_exact(golden):
    for name in MESHES:
        dom, res = golden[name + "_domain"], golden[name + "_res"]
        coeffs, P = golden[name + "_coeffs"], golden[name + "_P"]
        phi, grad = emu.interpolate(dom, res, coeffs, P, grad=True)
        np.testing.assert_array_equal(phi, golden[name + "_phi"])
        inside = golden[name + "_phi"] != DBL_MAX
        np.testing.assert_array_equal(grad[inside], golden[name + "_grad"][inside])
        assert (grad[~inside] == 0).all()
        np.testing.assert_array_equal(emu.interpolate(dom, res, coeffs, P), golden[name + "_phi"])
        # table mode == closed-form mode on an unreduced field
        cells = T.oracle_cell_table(res)
        cmap = np.arange(len(cells), dtype=np.uint32)
        np.testing.assert_array_equal(emu.interpolate(dom, res, coeffs, P, cells=cells, cell_map=cmap),
                                      golden[name + "_phi"])
        # removed cells and DBL_MAX coefficients
        cmap2 = cmap.copy()
        cmap2[::3] = 0xFFFFFFFF
        c2 = coeffs.copy()
        c2[::7] = DBL_MAX
        a, ga = emu.interpolate(dom, res, c2, P, grad=True, cells=cells, cell_map=cmap2)
        b, gb = T.oracle_interpolate(dom, res, c2, P, grad=True, cells=cells, cell_map=cmap2)
        np.testing.assert_array_equal(a, b)
        ok = b != DBL_MAX
        np.testing.assert_array_equal(ga[ok], gb[ok])


def test_interpolate_row_kernel_data_path_bit_exact(golden):
    """k_interpolate_rows on the host (emu_interpolate_rows): the cell-major copy as k_expand_cells builds it, waves of
    64 queries whose rows are staged with the kernel's lane map, and the two halves of interpolate_point_mode the
    kernel evaluates with (locate_query / evaluate_cell).  Same bits as the one-piece body and as the reference's
    golden values: unreduced, table mode, removed cells, DBL_MAX coefficients, batches that are no multiple of 64,
    queries outside the domain."""
    for name in MESHES:
        dom, res = golden[name + "_domain"], golden[name + "_res"]
        coeffs, P = golden[name + "_coeffs"], golden[name + "_P"]
        phi, grad = emu.interpolate_rows(dom, res, coeffs, P, grad=True)
        np.testing.assert_array_equal(phi, golden[name + "_phi"])
        inside = golden[name + "_phi"] != DBL_MAX
        assert (~inside).any() or name != "box"
        np.testing.assert_array_equal(grad[inside], golden[name + "_grad"][inside])
        assert (grad[~inside] == 0).all()
        for m in (1, 63, 64, 65, 1001):
            np.testing.assert_array_equal(emu.interpolate_rows(dom, res, coeffs, P[:m]), golden[name + "_phi"][:m])
        cells = T.oracle_cell_table(res)
        cmap = np.arange(len(cells), dtype=np.uint32)
        cmap2 = cmap.copy()
        cmap2[::3] = 0xFFFFFFFF
        c2 = coeffs.copy()
        c2[::7] = DBL_MAX
        keep = cmap2 != 0xFFFFFFFF
        cells2 = cells[keep]                                   # a reduced table: rows renumbered through the map
        cmap3 = cmap2.copy()
        cmap3[keep] = np.arange(keep.sum(), dtype=np.uint32)
        a, ga = emu.interpolate_rows(dom, res, c2, P, grad=True, cells=cells2, cell_map=cmap3)
        b, gb = emu.interpolate(dom, res, c2, P, grad=True, cells=cells2, cell_map=cmap3)
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(ga, gb)
        w, gw = T.oracle_interpolate(dom, res, c2, P, grad=True, cells=cells, cell_map=cmap2)
        np.testing.assert_array_equal(a, w)
        ok = w != DBL_MAX
        np.testing.assert_array_equal(ga[ok], gw[ok])


def test_config3_planes_bit_exact():
    """Icosphere nu=71 (100 820 triangles) at 256^3: four V planes through the costly centre
    and two X planes, against the oracle (bit-exact)."""
    V, F = T.icosphere(71)
    dom = T.oracle_default_domain(V)
    res = [256] * 3
    om, em = T.OracleMesh(V, F), emu.EmuMesh(V, F)
    plane = 257 * 257
    for b, e in ((127 * plane, 128 * plane + 1000), (257 ** 3 + 2 * 256 * 257 * 100, 257 ** 3 + 2 * 256 * 257 * 100 + 70000)):
        np.testing.assert_array_equal(em.sample_range(dom, res, b, e), om.sample_nodes(dom, res, b, e))


# ---- the single-point host evaluator of the product (dg_host_query.h) ------------------------------------
@pytest.mark.parametrize("mesh", ["box", "ico8", "torus", "bunny"])
def test_host_point_query_matches_oracle_bitwise(mesh):
    """dg_signed_distance_point's walk (near-first DFS over the product's BVH, conservative float
    bounds, reference arithmetic per triangle) gives the oracle's signed distance bit for bit; where the
    triangle differs the two triangles are exactly tied (nearest point and d^2 identical)."""
    V, F = {"box": T.box_mesh, "ico8": lambda: T.icosphere(8), "torus": T.torus, "bunny": T.bunny_mesh}[mesh]()
    rng = np.random.default_rng(11)
    lo, hi = V.min(0), V.max(0)
    P = rng.uniform(lo - 0.3 * (hi - lo), hi + 0.3 * (hi - lo), size=(6000, 3))
    P[:200] = V[rng.integers(0, len(V), 200)]                      # on vertices: many exact ties
    P[200:400] = 0.5 * (V[F[rng.integers(0, len(F), 200), 0]] + V[F[rng.integers(0, len(F), 200), 1]])
    P[400] = 0.5 * (lo + hi)
    om = T.OracleMesh(V, F)
    wd, wtri, went, wnear = om.signed_distance(P, full=True)
    em = emu.EmuMesh(V, F)
    d, tri, ent, near = em.host_signed_distance(P, full=True, threads=8)
    same = tri == wtri
    # the distance is the minimum of bit-exact d^2 values: identical whatever triangle attains it
    assert np.array_equal(d, wd)
    assert np.array_equal(ent[same], went[same])
    assert np.array_equal(near[same], wnear[same])
    # a different triangle must be an exact tie: the oracle's d^2 of both triangles is the same double
    T.assert_exact_ties(V, F, P, tri, wtri)


def test_host_point_query_concurrent_callers():
    """64 threads query one mesh at once (the reference declares signed_distance const + thread safe,
    TriangleMeshDistance.h:188,199): results equal the single-threaded ones."""
    V, F = T.icosphere(12)
    rng = np.random.default_rng(5)
    P = rng.uniform(-1.4, 1.4, size=(20000, 3))
    em = emu.EmuMesh(V, F)
    one = em.host_signed_distance(P, full=True, threads=1)
    many = em.host_signed_distance(P, full=True, threads=64)
    for a, b in zip(one, many):
        assert np.array_equal(a, b)


def test_host_point_query_nan_point():
    V, F = T.box_mesh()
    em = emu.EmuMesh(V, F)
    d, tri, ent, _ = em.host_signed_distance(np.array([[np.nan, 0.0, 0.0], [0.0, 0.0, 0.0]]), full=True)
    assert d[0] == np.finfo(np.float64).max and tri[0] == -1 and d[1] == -1.0


# ---- filtered K1 kernel (k_sample_fast): float filter + per-lane candidates + in-wave exact fallback ------------
def test_filtered_kernel_equals_exact_kernel_and_counts_its_paths():
    """Same bits with and without the filtered kernel, on lattices (heavy bricks parked with seed bounds
    included) and on point batches; the statistics show that the float filter really carried the work."""
    V, F = T.icosphere(12)
    dom = T.oracle_default_domain(V)
    res = [21, 18, 24]
    em = emu.EmuMesh(V, F)
    try:
        emu.set_fast(0)
        a = em.sample_range(dom, res)
        emu.set_fast(1)
        b = em.sample_range(dom, res)
        st = emu.fast_stats()
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(b, T.OracleMesh(V, F).sample_nodes(dom, res))
        assert st["bricks"] > 0 and st["tri_pairs"] > 0 and st["lanes"] > 0.9 * len(a) * 0.9
        assert st["sum_list"] >= st["lanes"]          # every lane ends with at least one candidate
        emu.set_heavy(256, 3)                          # almost every brick is parked: seeds, not triangles
        emu.set_fast(1)
        c = em.sample_range(dom, res)
        assert emu.fast_stats()["parked"] > 0
        np.testing.assert_array_equal(a, c)
    finally:
        emu.set_heavy()
        emu.set_fast(1)


def _apex_lattice(valence=40, res=(24, 24, 24), half=0.02):
    """a valence-`valence` bipyramid and a dense lattice around its upper apex"""
    V, F = T.bipyramid(valence)
    apex = V[valence]
    dom = np.concatenate([apex - half, apex + half])
    return V, F, dom, list(res)


def test_pooled_epilogue_of_the_filtered_kernel_both_branches():
    """The epilogue of k_sample_fast (dg_kernels_k1.hip, round 5): one candidate per lane tested by its owner, the tails of the lists
    pooled as (owner, triangle) pairs in list order and tested 64 per round, or -- when the pool does not hold them -- lane by lane.
    The emulator models the pool's index arithmetic (`before` from bit planes of the tail lengths, T, items, results handed back in
    list order) as the device computes it; on a valence-40 apex with a dense lattice BOTH branches occur by themselves (64 lanes with up
    to nine candidates exceed the 352 pairs the pool holds), a pool capped at 8 pairs sends nearly every wave down the second branch,
    a cap of 0 all of them: every variant the bits of the oracle, sign included."""
    V, F, dom, res = _apex_lattice()
    want = T.OracleMesh(V, F).sample_nodes(dom, res)
    em = emu.EmuMesh(V, F)
    try:
        seen = {}
        for cap in (0x7fffffff, 8, 0):
            emu.set_pool_cap(cap)
            emu.set_fast(1)
            got = em.sample_range(dom, res)
            np.testing.assert_array_equal(got, want, err_msg="pool_cap=%d" % cap)
            seen[cap] = emu.pool_stats()
            st = emu.fast_stats()
            assert max(k for k, c in enumerate(st["hist"]) if c) >= 6      # long lists exist: the lanes around the apex
        pooled, unpooled = seen[0x7fffffff]
        assert pooled > 0 and unpooled > 0, seen                          # the natural overflow (T > 352) happens on this lattice
        assert seen[8][1] > seen[0x7fffffff][1] and seen[0][0] == 0 and seen[0][1] == pooled + unpooled, seen
        # ... and on an ordinary mesh everything pools
        V2, F2 = T.icosphere(12)
        d2 = T.oracle_default_domain(V2)
        emu.set_pool_cap()
        emu.set_fast(1)
        np.testing.assert_array_equal(emu.EmuMesh(V2, F2).sample_range(d2, [21, 18, 24]), T.OracleMesh(V2, F2).sample_nodes(d2, [21, 18, 24]))
        assert emu.pool_stats()[0] > 0 and emu.pool_stats()[1] == 0
    finally:
        emu.set_pool_cap()
        emu.set_fast(1)


def test_filtered_kernel_fallbacks_degenerate_triangles_and_far_points():
    """What the float filter cannot serve gets the exact traversal in the same wave: degenerate triangles
    (zero area, duplicated vertices), points whose coordinates leave the filter's range (1e20) next to
    ordinary points, NaN points."""
    V, F = T.icosphere(5)
    V = np.vstack([V, V[3] + [0, 0, 0.25], V[3] + [0, 0, 0.5]])
    n = len(V)
    F2 = np.vstack([F, [[3, n - 2, n - 1]], [[7, 7, 9]], [[n - 1, n - 2, 3]]]).astype(np.uint32)  # collinear, duplicate vertex
    om, em = T.OracleMesh(V, F2), emu.EmuMesh(V, F2)
    rng = np.random.default_rng(9)
    P = rng.uniform(-1.6, 1.6, size=(1500, 3))
    P[::97] *= 1.0e20
    emu.set_fast(1)
    got = em.signed_distance(P)
    st = emu.fast_stats()
    want = om.signed_distance(P)
    np.testing.assert_array_equal(np.abs(got), np.abs(want))
    assert st["redo_bricks"] > 0                       # the fallback ran
    dom = T.oracle_default_domain(V)
    emu.set_fast(1)
    np.testing.assert_array_equal(np.abs(em.sample_range(dom, [9, 11, 8])), np.abs(om.sample_nodes(dom, [9, 11, 8])))
    assert emu.fast_stats()["redo_bricks"] > 0
    Q = P[:130].copy()
    Q[5] = np.nan
    emu.set_fast(1)
    d = em.signed_distance(Q)
    assert d[5] == DBL_MAX
    np.testing.assert_array_equal(np.abs(np.delete(d, 5)), np.abs(np.delete(want[:130], 5)))


def test_float_filter_interval_contains_the_double_value():
    """The error analysis of dg_geom.h on random and adversarial inputs: needles, slivers, points near
    sides / vertices, far points, large offsets from the origin -- checked with the product's own code
    through the emulator (a violated interval would make the filtered kernel drop the true winner)."""
    rng = np.random.default_rng(21)
    worst = emu.filter_interval_check(rng, n_tri=4000, n_pts=64)
    assert worst["violations"] == 0, worst
    assert worst["checked"] > 200000


def test_tile_major_copy_is_the_same_field(golden):
    """The tile-major layout (4^3-cell tiles, 736 doubles each; dg_lattice.h) addresses exactly the reference's
    nodes: K2 and K3 bodies give the same bits through it, on resolutions that are no multiples of 4 too."""
    rng = np.random.default_rng(3)
    try:
        for res in ([4, 4, 4], [5, 7, 3], [9, 2, 6], [1, 1, 1]):
            dom = np.array([-1.0, -0.5, 0.0, 1.5, 1.0, 2.0])
            coeffs = rng.normal(size=T.n_nodes(res))
            P = rng.uniform(dom[:3] - 0.1, dom[3:] + 0.1, size=(3000, 3))
            emu.set_tile_major(0)
            a, ga = emu.interpolate(dom, res, coeffs, P, grad=True)
            emu.set_tile_major(1)
            b, gb = emu.interpolate(dom, res, coeffs, P, grad=True)
            np.testing.assert_array_equal(a, b)
            np.testing.assert_array_equal(ga, gb)
            wa, wg = T.oracle_interpolate(dom, res, coeffs, P, grad=True)
            np.testing.assert_array_equal(b, wa)
            inside = b != DBL_MAX   # (the oracle leaves the gradient of "no value" queries untouched)
            np.testing.assert_array_equal(gb[inside], wg[inside])
        dom, res = golden["ico8_domain"], golden["ico8_res"]
        coeffs = golden["ico8_coeffs"]
        emu.set_tile_major(0)
        d0 = emu.density_map(dom, res, coeffs, 0.1, 1000.0, begin=0, end=600)
        emu.set_tile_major(1)
        d1 = emu.density_map(dom, res, coeffs, 0.1, 1000.0, begin=0, end=600)
        np.testing.assert_array_equal(d0, d1)
        # "no value" coefficients: the tile copy answers with one flag bit per cell (kTmFlags) instead of 32 compares
        c2 = coeffs.copy()
        c2[::41] = DBL_MAX
        emu.set_tile_major(0)
        e0 = emu.density_map(dom, res, c2, 0.1, 1000.0, begin=0, end=600)
        emu.set_tile_major(1)
        e1 = emu.density_map(dom, res, c2, 0.1, 1000.0, begin=0, end=600)
        np.testing.assert_array_equal(e0, e1)
        assert (e0 != d0).any()
    finally:
        emu.set_tile_major(0)


def test_xmajor_copy_is_the_same_field(golden):
    """The x-major copy of the Y and Z edge classes (dg_lattice.h: K3's row-block kernel reads all 16 coefficient pairs of a
    cell from rows along x) addresses exactly the reference's nodes: same bits from K2's and K3's bodies, "no value"
    coefficients answered by its one bit per cell."""
    rng = np.random.default_rng(31)
    try:
        for res in ([4, 4, 4], [5, 7, 3], [9, 2, 6], [1, 1, 1], [70, 2, 3]):
            dom = np.array([-1.0, -0.5, 0.0, 1.5, 1.0, 2.0])
            coeffs = rng.normal(size=T.n_nodes(res))
            P = rng.uniform(dom[:3] - 0.1, dom[3:] + 0.1, size=(3000, 3))
            emu.set_tile_major(0)
            a, ga = emu.interpolate(dom, res, coeffs, P, grad=True)
            emu.set_tile_major(2)
            b, gb = emu.interpolate(dom, res, coeffs, P, grad=True)
            np.testing.assert_array_equal(a, b)
            np.testing.assert_array_equal(ga, gb)
        dom, res = golden["ico8_domain"], golden["ico8_res"]
        coeffs = golden["ico8_coeffs"]
        emu.set_tile_major(0)
        d0 = emu.density_map(dom, res, coeffs, 0.1, 1000.0, begin=0, end=600)
        emu.set_tile_major(2)
        d1 = emu.density_map(dom, res, coeffs, 0.1, 1000.0, begin=0, end=600)
        np.testing.assert_array_equal(d0, d1)
        c2 = coeffs.copy()
        c2[::41] = DBL_MAX
        emu.set_tile_major(0)
        e0 = emu.density_map(dom, res, c2, 0.1, 1000.0, begin=0, end=600)
        emu.set_tile_major(2)
        e1 = emu.density_map(dom, res, c2, 0.1, 1000.0, begin=0, end=600)
        np.testing.assert_array_equal(e0, e1)
        assert (e0 != d0).any()
    finally:
        emu.set_tile_major(0)


def test_point_lanes_with_seven_nodes_equal_the_per_node_body(golden):
    """k_density_cells (dg_density_cells.h): a lane owns a lattice point with its seven nodes, evaluates the points that share a
    cell from one fetch and takes the axis states of the shifted coordinates from tables.  Every node of the lattice is written
    exactly once, and with the bits of the per-node body (density_prefilter + density_integral) -- on lattices that are no
    multiples of the wave shape, support radii from a fraction of a cell to several cells (the seven points then spread over
    one to four cells), with and without the node predicate, node ranges, masks, and fields spoilt with "no value", NaN and
    Inf coefficients (the latter evaluate every one of the 4096 points)."""
    rng = np.random.default_rng(77)
    dom, res = golden["ico8_domain"], [int(r) for r in golden["ico8_res"]]
    coeffs = golden["ico8_coeffs"]
    n = T.n_nodes(res)
    cases = [(dom, res, coeffs, 0.1, True), (dom, res, coeffs, 0.23, False)]
    V, F = T.icosphere(4)
    d2 = T.oracle_default_domain(V)
    for r2, h in (([5, 7, 3], 0.3), ([17, 2, 4], 0.08), ([1, 1, 1], 0.5), ([3, 18, 2], 0.6)):
        c2 = T.OracleMesh(V, F).sample_nodes(d2, r2)
        cases.append((d2, r2, c2, h, True))
        cases.append((d2, r2, c2, h, False))
    emu.set_tile_major(2)
    try:
        for ci, (dm, rs, cf, h, band) in enumerate(cases):
            nn = T.n_nodes(rs)
            want = emu.density_map(dm, rs, cf, h, 1000.0, band=band)
            got, hits = emu.density_cells(dm, rs, cf, h, 1000.0, band=band, block=(1 + ci % 3, 2, 5))
            assert (hits == 1).all(), (rs, np.unique(hits))
            np.testing.assert_array_equal(got, want, err_msg="%s h=%g band=%s" % (rs, h, band))
            assert ((got != 0.0) & (got != DBL_MAX)).sum() > 0
            # a node range (lanes outside idle) and a mask
            b, e = nn // 3, nn - 2
            got_r, hits_r = emu.density_cells(dm, rs, cf, h, 1000.0, band=band, begin=b, end=e)
            assert (hits_r == 1).all()
            np.testing.assert_array_equal(got_r, want[b:e])
            mask = (np.arange(nn) % 5 != 0).astype(np.uint8)
            got_m, _ = emu.density_cells(dm, rs, cf, h, 1000.0, band=band, mask=mask)
            np.testing.assert_array_equal(got_m[mask == 1], want[mask == 1])
            assert (got_m[mask == 0] == DBL_MAX).all()
        # spoilt fields
        for rs, h in (([5, 7, 3], 0.3), ([9, 4, 6], 0.2)):
            cf = T.OracleMesh(V, F).sample_nodes(d2, rs)
            nn = T.n_nodes(rs)
            spoilt = cf.copy()
            spoilt[rng.integers(0, nn, size=6)] = DBL_MAX
            worse = spoilt.copy()
            worse[rng.integers(0, nn, size=2)] = np.nan
            worse[rng.integers(0, nn, size=2)] = np.inf
            for name, c in (("no value", spoilt), ("nan / inf", worse)):
                want = emu.density_map(d2, rs, c, h, 1000.0, band=True)
                got, hits = emu.density_cells(d2, rs, c, h, 1000.0, band=True)
                assert (hits == 1).all()
                np.testing.assert_array_equal(got, want, err_msg="%s %s" % (rs, name))
    finally:
        emu.set_tile_major(0)


def test_band_limited_copy_data_path_bit_exact(golden):
    """K2's band-limited cell-major copy (dg_lattice.h band_row_of(): one bit per cell row + a running count per 64 rows): the
    look-up finds exactly the rows the builder kept, mapped and unmapped queries evaluate to the bits of the plain path (and of
    the reference's golden values) -- wide, narrow, empty and all-inclusive bands, unreduced and table-mode fields with removed
    cells and "no value" coefficients, row counts that are no multiples of 64."""
    for name in ("torus", "box"):
        dom, res = golden[name + "_domain"], golden[name + "_res"]
        coeffs, P = golden[name + "_coeffs"], golden[name + "_P"]
        want_phi, want_grad = golden[name + "_phi"], golden[name + "_grad"]
        fin = coeffs[coeffs != DBL_MAX]
        n_cells = int(np.prod(res))
        seen = set()
        for lo, hi in ((-0.05, 0.05), (float(np.median(fin)), float(fin.max())), (float(fin.max()) + 1.0, float(fin.max()) + 2.0),
                       (float(fin.min()) - 1.0, float(fin.max()) + 1.0)):
            phi, grad, rows, mapped = emu.interpolate_band(dom, res, coeffs, lo, hi, P, grad=True)
            np.testing.assert_array_equal(phi, want_phi)
            inside = want_phi != DBL_MAX
            np.testing.assert_array_equal(grad[inside], want_grad[inside])
            assert 0 <= rows <= n_cells and mapped <= inside.sum()
            seen.add(rows)
        assert 0 in seen and n_cells in seen and len(seen) >= 3
        cells = T.oracle_cell_table(res)
        cmap = np.arange(len(cells), dtype=np.uint32)
        cmap[::3] = 0xFFFFFFFF
        c2 = coeffs.copy()
        c2[::7] = DBL_MAX
        a, ga = emu.interpolate(dom, res, c2, P, grad=True, cells=cells, cell_map=cmap)
        b, gb, rows, mapped = emu.interpolate_band(dom, res, c2, -0.1, 0.1, P, grad=True, cells=cells, cell_map=cmap)
        np.testing.assert_array_equal(a, b)
        np.testing.assert_array_equal(ga, gb)
        assert rows > 0


def test_division_free_quotient_by_the_support_radius_is_the_division():
    """k_density_cells forms gamma = 1 - d / h with a multiplication by RN(1 / h) and two residual corrections
    (dg_density_cells.h k3c_div_h(): Markstein's theorem).  Same bits as the division for 10^7 numerators per support
    radius -- random over the whole admitted range, around h, and next to rounding boundaries of the quotient."""
    rng = np.random.default_rng(9)
    hs = [0.1, 0.05, 0.2, 0.3, 1.0 / 3.0, 0.0123456789, 1.0, 2.5e-3, 7.0, 1.0e-12, 1.0e12] + list(rng.uniform(1e-3, 2.0, 9))
    for t, h in enumerate(hs):
        assert emu.div_h_mismatches(h, 10_000_000 if t < 4 else 2_000_000, 100 + t) == 0, h


def test_scalar_division_by_launch_constants_is_exact():
    """The brick map divides by launch constants with a host-made reciprocal (dg_kernels.h: udiv_by): exact
    quotient and remainder for every 32-bit dividend and every divisor >= 1, including the corners."""
    rng = np.random.default_rng(5)
    edge = np.array([0, 1, 2, 3, 5, 7, 63, 64, 65, 255, 256, 257, 641, 65535, 65536, 65537, 2**24 - 1, 2**24, 2**31 - 1,
                     2**31, 2**31 + 1, 2**32 - 2, 2**32 - 1], dtype=np.uint64)
    dd = np.concatenate([edge[1:], rng.integers(1, 5000, 200, dtype=np.uint64), rng.integers(1, 2**32, 200, dtype=np.uint64)])
    for d in dd:
        # dividends: the corners, random ones, and the neighbourhoods of multiples of d (where a quotient off by one shows)
        k = rng.integers(0, max(1, (2**32 - 1) // int(d)) + 1, 64).astype(np.int64)
        near = np.concatenate([k * int(d) + o for o in (-1, 0, 1)])
        near = near[(near >= 0) & (near < 2**32)]
        n = np.concatenate([edge.astype(np.int64), rng.integers(0, 2**32, 256), near]).astype(np.uint32)
        assert emu.udiv_mismatches(n, np.full(len(n), d, dtype=np.uint32)) == 0, int(d)


def test_blocked_brick_order_is_a_bijection():
    """K3 enumerates the bricks of a class in blocks of 4 x 4 x 8 (dg_kernels.h: map_lane with brick_blocking):
    every lattice node still exactly once, same values -- on lattices whose brick counts are no multiples of
    the block, on sub-ranges and on shards."""
    V, F = T.icosphere(3)
    dom = T.oracle_default_domain(V)
    em = emu.EmuMesh(V, F)
    try:
        for res in ([5, 5, 5], [18, 7, 35], [33, 20, 3], [1, 1, 40]):
            n = T.n_nodes(res)
            emu.set_brick_blocking(0)
            ref = em.sample_range(dom, res)
            emu.set_brick_blocking(1)
            got = em.sample_range(dom, res)
            assert (em.written == 1).all()
            np.testing.assert_array_equal(got, ref)
            b, e = n // 5, (3 * n) // 4 + 3
            part = em.sample_range(dom, res, b, e)
            assert (em.written == 1).all()
            np.testing.assert_array_equal(part, ref[b:e])
            sh = em.sample_shard(dom, res, 1, 3)
            assert (em.written == 1).all()
            emu.set_brick_blocking(0)
            np.testing.assert_array_equal(sh, em.sample_shard(dom, res, 1, 3))
    finally:
        emu.set_brick_blocking(0)
