"""TOTAL parity on the judged configurations (BASELINE.json configs[1..4]) against the UNMODIFIED
reference: tests/golden/lattice_digests.npz holds, per block of 2^20 consecutive nodes (queries), the first
16 bytes of SHA-256 over the raw doubles the reference produced (tests/golden/make_digests.py ran the
reference's own node loop / interpolate over everything, in the build container).  Here the GPU result is
reduced the same way and EVERY block is compared: 100 % of the lattice is pinned bit for bit, heavy bricks
included, and config 5 (10 M queries, value and gradient) at its stated size.

All through the C ABI (ctypes); the field stays on the device between K1 and K2/K3."""
import os

import numpy as np
import pytest

import dgtest as T

pytestmark = pytest.mark.gpu
DBL_MAX = np.finfo(np.float64).max
GOLD = os.path.join(T.GOLDEN, "lattice_digests.npz")


@pytest.fixture(scope="module")
def gold():
    assert os.path.exists(GOLD), "tests/golden/lattice_digests.npz missing (python tests/golden/make_digests.py)"
    return dict(np.load(GOLD))


@pytest.fixture(scope="module")
def dg():
    import discregrid_amd
    discregrid_amd.load_library()
    assert discregrid_amd.device_count() >= 1
    discregrid_amd.set_device(0)
    return discregrid_amd


def mismatching_blocks(got, want):
    assert got.shape == want.shape, (got.shape, want.shape)
    return np.flatnonzero((got != want).any(axis=1))


def sample_on_device(dg, torch, V, F, dom, res):
    grid = dg.grid_desc(dom[:3], dom[3:], res)
    n = dg.n_nodes(grid)
    mesh = dg.Mesh(V, F)
    field = torch.empty(n, dtype=torch.float64, device="cuda")
    mesh.sample_nodes_device(grid, 0, n, field.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return mesh, grid, field


def test_config2_bunny_128_every_node(dg, gold):
    """BASELINE configs[1]: bunny, 128^3, all 14 926 977 coefficients == the reference's."""
    import torch
    V, F = T.bunny_mesh()
    dom = gold["bunny128_domain"]
    np.testing.assert_array_equal(dom, dg.default_domain(V))
    _, _, field = sample_on_device(dg, torch, V, F, dom, [128] * 3)
    assert len(field) == int(gold["bunny128_nodes"])
    bad = mismatching_blocks(T.block_digests(field.cpu().numpy()), gold["bunny128_digest"])
    assert len(bad) == 0, "blocks of 2^20 nodes that differ from the reference: %s" % bad[:10]


def test_dragon_128_every_node(dg, gold):
    """The reference's third sample mesh (cmd/generate_sdf/resources/dragon.obj, 79 988 triangles -- thin features, a mesh
    that is neither smooth like the bunny nor regular like the icosphere), 128^3: all 14 926 977 coefficients == the
    reference's."""
    import torch
    V, F = T.dragon_mesh()
    assert len(F) == 79988
    dom = gold["dragon128_domain"]
    np.testing.assert_array_equal(dom, dg.default_domain(V))
    _, _, field = sample_on_device(dg, torch, V, F, dom, [128] * 3)
    assert len(field) == int(gold["dragon128_nodes"])
    bad = mismatching_blocks(T.block_digests(field.cpu().numpy()), gold["dragon128_digest"])
    assert len(bad) == 0, "blocks of 2^20 nodes that differ from the reference: %s" % bad[:10]


def test_one_million_triangles_strided_nodes(dg):
    """An icosphere of 1 003 520 triangles (nu = 224; ten times the judged mesh: a deeper tree, 3 % of the triangles per
    brick neighbourhood), 128^3: every 746-th lattice node == what the UNMODIFIED reference's node loop returned for it
    (tests/golden/big_mesh_sample.npz, written by tests/golden/make_digests.py ico224_sample)."""
    import torch
    z = np.load(os.path.join(T.GOLDEN, "big_mesh_sample.npz"))
    V, F = T.icosphere(int(z["nu"]))
    assert len(F) == int(z["triangles"]) == 1003520
    dom = z["domain"]
    np.testing.assert_array_equal(dom, dg.default_domain(V))
    mesh, _, field = sample_on_device(dg, torch, V, F, dom, [int(r) for r in z["res"]])
    got = field.cpu().numpy()[z["idx"].astype(np.int64)]
    np.testing.assert_array_equal(got, z["sd"])
    assert mesh.info()["n_triangles"] == 1003520


@pytest.fixture(scope="module")
def ico256(dg, gold):
    import torch
    V, F = T.icosphere(71)
    dom = gold["ico71_256_domain"]
    np.testing.assert_array_equal(dom, dg.default_domain(V))
    mesh, grid, field = sample_on_device(dg, torch, V, F, dom, [256] * 3)
    return V, F, dom, mesh, grid, field


def test_config3_icosphere_256_every_node(dg, gold, ico256):
    """BASELINE configs[2] (the headline workload): all 118 425 857 coefficients == the reference's,
    the 381 heavy bricks around the sphere's centre included."""
    V, F, dom, mesh, grid, field = ico256
    assert len(field) == int(gold["ico71_256_nodes"])
    heavy, split = mesh.last_heavy_bricks()
    assert heavy > 0 and split > 0            # the split path ran in this very launch
    bad = mismatching_blocks(T.block_digests(field.cpu().numpy()), gold["ico71_256_digest"])
    assert len(bad) == 0, "blocks of 2^20 nodes that differ from the reference: %s" % bad[:10]


def _config5_points(gold, dom, evaluate_stream):
    """The two query sets of config 5.  The shell is selected with the values `evaluate_stream` returns for the
    26 M-point stream, so the selection itself is part of the check."""
    nq = int(gold["c5_n_queries"])
    P = T.uniform_points(1234, nq, dom[:3], dom[3:])
    C = T.uniform_points(4321, int(gold["c5_shell_stream"]), dom[:3], dom[3:])
    pc = evaluate_stream(C)
    keep = np.flatnonzero((pc != DBL_MAX) & (np.abs(pc) < 0.2))
    assert len(keep) >= nq and keep[nq - 1] == int(gold["c5_shell_last_candidate"])
    S = np.ascontiguousarray(C[keep[:nq]])
    assert len(mismatching_blocks(T.block_digests(S), gold["c5_shell_points"])) == 0
    return P, S


def _check_config5(gold, evaluate, P, S, what):
    phi, phi_g, grad = evaluate(P)
    np.testing.assert_array_equal(phi[:64], gold["c5_uniform_phi_head"])
    for name, got in (("phi", phi), ("phi_g", phi_g), ("grad", grad)):
        bad = mismatching_blocks(T.block_digests(got), gold["c5_uniform_" + name])
        assert len(bad) == 0, (what, "uniform", name, bad[:10])
    phi, phi_g, grad = evaluate(S)
    for name, got in (("phi", phi), ("phi_g", phi_g), ("grad", grad)):
        bad = mismatching_blocks(T.block_digests(got), gold["c5_shell_" + name])
        assert len(bad) == 0, (what, "shell", name, bad[:10])


def _device_evaluator(torch, fld, order=None):
    """P -> (phi, phi with gradient, gradient) through dg_interpolate_batch_device; with `order` (a callable
    P -> permutation) the queries are handed over in that order and the results put back before digesting."""
    s = torch.cuda.current_stream().cuda_stream

    def evaluate(P):
        perm = None if order is None else order(P)
        Q = P if perm is None else np.ascontiguousarray(P[perm])
        d_P = torch.from_numpy(Q).cuda()
        phi = torch.empty(len(Q), dtype=torch.float64, device="cuda")
        fld.interpolate_device(d_P.data_ptr(), len(Q), phi.data_ptr(), stream=s)
        phi_g = torch.empty(len(Q), dtype=torch.float64, device="cuda")
        grad = torch.empty(3 * len(Q), dtype=torch.float64, device="cuda")
        fld.interpolate_device(d_P.data_ptr(), len(Q), phi_g.data_ptr(), grad.data_ptr(), stream=s)
        torch.cuda.synchronize()
        out = [phi.cpu().numpy(), phi_g.cpu().numpy(), grad.cpu().numpy().reshape(-1, 3)]
        if perm is not None:
            for i, a in enumerate(out):
                b = np.empty_like(a)
                b[perm] = a
                out[i] = b
        return out
    return evaluate


@pytest.fixture(scope="module")
def config5_points(dg, gold, ico256):
    import torch
    V, F, dom, mesh, grid, field = ico256
    fld = dg.Field(grid, d_coeffs=field.data_ptr(), n_coeffs=len(field))
    P, S = _config5_points(gold, dom, lambda C: _stream_values(torch, fld, C))
    fld.close()
    return P, S


def _stream_values(torch, fld, C):
    d_C = torch.from_numpy(C).cuda()
    phic = torch.empty(len(C), dtype=torch.float64, device="cuda")
    fld.interpolate_device(d_C.data_ptr(), len(C), phic.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return phic.cpu().numpy()


def test_config5_interpolate_10m_queries(dg, gold, ico256, config5_points):
    """BASELINE configs[4], part 1: 10 M uniform queries (std::mt19937_64 seed 1234) and 10 M queries of the
    SPH-like shell |phi| < 2h on the 256^3 field, value-only and value + gradient: every block of 2^20
    results == the reference's CubicLagrangeDiscreteGrid::interpolate (:977-1063).  Attached device array,
    plain layout: the binned gather kernels."""
    import torch
    V, F, dom, mesh, grid, field = ico256
    P, S = config5_points
    fld = dg.Field(grid, d_coeffs=field.data_ptr(), n_coeffs=len(field))
    _check_config5(gold, _device_evaluator(torch, fld), P, S, "attached, plain layout, binned")
    fld.close()


def cell_order(dom, res):
    """P -> the permutation that sorts the queries by the cell they fall into (the order an SPH code's
    spatially sorted particles arrive in): K2's "already ordered" branch."""
    lo, hi, res = np.asarray(dom[:3]), np.asarray(dom[3:]), np.asarray(res)

    def order(P):
        c = np.clip(((P - lo) / (hi - lo) * res).astype(np.int64), 0, res - 1)
        return np.argsort(c[:, 0] + res[0] * (c[:, 1] + res[1] * c[:, 2]), kind="stable")
    return order


@pytest.mark.parametrize("path", ["cell_major_rows", "cell_major_per_lane", "owned_auto_copy", "tile_major",
                                  "cell_sorted_input", "cell_sorted_input_cell_major", "no_binning", "band_copy", "band_copy_thin",
                                  "radix_sorted_gather", "tiles_plain_item_order"])
def test_config5_interpolate_every_k2_path(dg, gold, ico256, config5_points, path, monkeypatch):
    """The same 2 x 3 x 10 digest blocks through EVERY other K2 path that carries a quoted number
    (secondary.k2_interpolate on the bench line): the cooperative row kernel on the cell-major copy
    (k_interpolate_rows: the 18 Gq/s figure), the per-lane kernels on that copy, an OWNED field that builds the
    copy by itself on its first large batch, the tile-major copy, cell-sorted input (the device-side "already
    ordered" decision; results un-permuted before digesting) on both layouts, the unbinned gather, (round 6) the radix-sorted
    per-lane gather that was the default until round 5 and the staged gather without its XCD-aware item order, and (round 4) the
    band-limited cell-major copy: rows for the cells that reach into |phi| <= 2h + cell diagonal (every shell query takes the
    row path, four uniform queries in five the gather, in one launch) and for a band so thin that most shell queries
    straddle it or miss it."""
    import torch
    V, F, dom, mesh, grid, field = ico256
    P, S = config5_points
    order = None
    if path == "owned_auto_copy":
        fld = dg.Field(grid, field.cpu().numpy())          # dg_field_create: the library owns the coefficients
    else:
        fld = dg.Field(grid, d_coeffs=field.data_ptr(), n_coeffs=len(field))
    if path in ("cell_major_rows", "cell_major_per_lane", "cell_sorted_input_cell_major"):
        fld.build_cell_major(stream=torch.cuda.current_stream().cuda_stream)
    if path == "cell_major_per_lane":
        T.force(monkeypatch, k2_rows=0)
    if path == "tile_major":
        fld.build_tile_major(stream=torch.cuda.current_stream().cuda_stream)
    if path.startswith("cell_sorted_input"):
        order = cell_order(dom, [256] * 3)
    if path == "no_binning":
        T.force(monkeypatch, k2_binning=0)
    if path == "radix_sorted_gather":      # the default of rounds 1-5: radix sort by cell, per-lane gather
        T.force(monkeypatch, k2_tiles=0)
    if path == "tiles_plain_item_order":   # round 6's staged gather without the XCD-aware order of its work items
        T.force(monkeypatch, k2_tile_chunk=0)
    if path.startswith("band_copy"):
        n_cells = 256 ** 3
        diag = float(np.linalg.norm((np.asarray(dom[3:]) - np.asarray(dom[:3])) / 256.0))
        band = (2 * 0.1 + diag) if path == "band_copy" else 0.02
        rows = fld.build_cell_major_band(-band, band, stream=torch.cuda.current_stream().cuda_stream)
        info = fld.info()
        print("%s: |phi| <= %.4f -> %d of %d cell rows (%.1f %%), copy + map %.2f GB against %.2f GB of field"
              % (path, band, rows, n_cells, 100.0 * rows / n_cells, (rows * 256 + n_cells * 4) / 1e9, len(field) * 8 / 1e9))
        # (h = 0.1 on a unit sphere in a box that hugs it: the shell is THICK -- 57 % of the cells; a band of +-0.02: a few per cent)
        assert info["band_rows"] == rows and 0 < rows < (0.65 if path == "band_copy" else 0.12) * n_cells
    _check_config5(gold, _device_evaluator(torch, fld, order), P, S, path)
    if path == "owned_auto_copy":
        assert fld.has_cell_major(), "the owned field did not build its cell-major copy on a 10 M batch"
    fld.close()


def _density_on_device(dg, torch, grid, field, h, rho0):
    s = torch.cuda.current_stream().cuda_stream
    fld = dg.Field(grid, d_coeffs=field.data_ptr(), n_coeffs=len(field))
    dens = torch.empty(len(field), dtype=torch.float64, device="cuda")
    fld.density_map_nodes_device(h, rho0, True, 0, len(field), dens.data_ptr(), stream=s)
    torch.cuda.synchronize()
    got = dens.cpu().numpy()
    fld.close()
    return got


def test_density_map_128_every_node(dg, gold):
    """K3 at FULL lattice against the UNMODIFIED reference: the 128^3 icosphere SDF (its own digests checked
    first) through GenerateDensityMap's node predicate + density_func (cmd/generate_density_map/main.cpp:86-133;
    11.3 M integrated nodes = 46 G reference interpolate calls): every block of 2^20 of the 14 926 977 results."""
    import torch
    if "density128_digest" not in gold:
        pytest.skip("density128 digests not generated (python tests/golden/make_digests.py density128)")
    V, F = T.icosphere(71)
    dom = gold["ico71_128_domain"]
    mesh, grid, field = sample_on_device(dg, torch, V, F, dom, [128] * 3)
    assert len(field) == int(gold["ico71_128_nodes"])
    assert len(mismatching_blocks(T.block_digests(field.cpu().numpy()), gold["ico71_128_digest"])) == 0
    got = _density_on_device(dg, torch, grid, field, float(gold["density128_h"]), float(gold["density128_rho0"]))
    stride = int(gold["density128_sample_stride"])
    np.testing.assert_array_equal(got[::stride], gold["density128_sample"])     # readable failure first
    bad = mismatching_blocks(T.block_digests(got), gold["density128_digest"])
    assert len(bad) == 0, "blocks of 2^20 nodes that differ from the reference: %s" % bad[:10]
    integrated = int(np.count_nonzero((got != DBL_MAX) & (field.cpu().numpy() <= 0.2)))
    assert integrated == int(gold["density128_integrated_nodes"])


def test_config5_density_map_256(dg, gold, ico256):
    """BASELINE configs[4], part 2: GenerateDensityMap's node function (K3) on the whole 256^3 SDF.  With the
    reference digests (tests/golden/make_digests.py density256: the unmodified reference over all 118 425 857
    nodes, 367 G interpolate calls) every block of 2^20 results is compared; without them a strided sample of
    integrated, zero and rejected nodes == the oracle's restatement of cmd/generate_density_map/main.cpp:86-133."""
    import torch
    V, F, dom, mesh, grid, field = ico256
    got = _density_on_device(dg, torch, grid, field, 0.1, 1000.0)
    if "density256_digest" in gold:
        stride = int(gold["density256_sample_stride"])
        np.testing.assert_array_equal(got[::stride], gold["density256_sample"])
        bad = mismatching_blocks(T.block_digests(got), gold["density256_digest"])
        assert len(bad) == 0, "blocks of 2^20 nodes that differ from the reference: %s" % bad[:10]
        return
    coeffs = field.cpu().numpy()
    integrated = np.flatnonzero((got != DBL_MAX) & (got != 0.0))
    zero = np.flatnonzero(got == 0.0)
    rejected = np.flatnonzero(got == DBL_MAX)
    assert len(integrated) > 10_000_000 and len(zero) > 0 and len(rejected) > 0
    pick = np.concatenate([integrated[:: max(1, len(integrated) // 300)], zero[:: max(1, len(zero) // 40)],
                           rejected[:: max(1, len(rejected) // 40)]])
    for l in pick:
        want = T.oracle_density_map(dom, [256] * 3, coeffs, 0.1, 1000.0, band=True, begin=int(l), end=int(l) + 1)
        assert want[0] == got[l] or (np.isnan(want[0]) and np.isnan(got[l])), (int(l), want[0], got[l])


def test_config4_lattice_512_every_node(dg, gold):
    """BASELINE configs[3]'s lattice (same mesh, 512^3, 943 460 865 nodes = 7.5 GB): sampled (a) by one
    launch and (b) as the 8 shards of the multi-GPU deal, computed in turn on this one GPU and unpacked --
    both == the reference's coefficients in every block of 2^20 nodes."""
    import torch
    if "ico71_512_digest" not in gold:
        pytest.skip("512^3 digests not generated")
    V, F = T.icosphere(71)
    dom = gold["ico71_512_domain"]
    mesh, grid, field = sample_on_device(dg, torch, V, F, dom, [512] * 3)
    assert len(field) == int(gold["ico71_512_nodes"])
    bad = mismatching_blocks(T.block_digests(field.cpu().numpy()), gold["ico71_512_digest"])
    assert len(bad) == 0, "direct launch: blocks that differ from the reference: %s" % bad[:10]
    s = torch.cuda.current_stream().cuda_stream
    nr = 8
    _, stride = dg.shard_layout(grid, 0, nr)
    gathered = torch.zeros(nr * stride, dtype=torch.float64, device="cuda")
    for r in range(nr):
        mesh.sample_shard_device(grid, r, nr, gathered[r * stride:].data_ptr(), stream=s)
    field.fill_(0.0)
    dg.unpack_shards_device(grid, nr, gathered.data_ptr(), stride, field.data_ptr(), stream=s)
    torch.cuda.synchronize()
    del gathered
    bad = mismatching_blocks(T.block_digests(field.cpu().numpy()), gold["ico71_512_digest"])
    assert len(bad) == 0, "8 shards + unpack: blocks that differ from the reference: %s" % bad[:10]
