"""K3 on the GPU through the C ABI and through the GenerateDensityMap tool, against the
reference's outputs (golden) and the oracle."""
import os
import subprocess
import time

import numpy as np
import pytest

import dgtest as T

pytestmark = pytest.mark.gpu
DBL_MAX = np.finfo(np.float64).max
BUILD = os.path.join(T.ROOT, "discregrid_amd", "cpp", "build")


@pytest.fixture(scope="module")
def dg():
    import discregrid_amd
    discregrid_amd.load_library()
    assert discregrid_amd.device_count() >= 1
    return discregrid_amd


@pytest.mark.parametrize("name,res,h,key", [("torus_9_14_6.cdf", [9, 14, 6], 0.15, "torus_density_h015"),
                                             ("torus_16_16_6.cdf", [16, 16, 6], 0.1, "torus16_density_h01")])
def test_density_map_vs_reference_golden(dg, golden, monkeypatch, name, res, h, key):
    g = T.read_cdf(os.path.join(T.GOLDEN, name))
    grid = dg.grid_desc(g["domain"][:3], g["domain"][3:], res)
    f = dg.Field(grid, g["nodes"][0])
    n = dg.n_nodes(grid)
    got = f.density_map_nodes(n, h, 1000.0, True)
    want = golden[key]
    rel = T.rel_err(got, want)
    print("%s: max rel err %.3e, %d / %d not bit-equal, K3 %.2f ms" % (key, rel, int((got != want).sum()), n,
                                                                       dg.last_kernel_ms()))
    assert rel <= 1e-10
    np.testing.assert_array_equal(got, want)
    # ranges, mask, cell-major layout, no predicate
    np.testing.assert_array_equal(f.density_map_nodes(n, h, 1000.0, True, 777, 2500), want[777:2500])
    mask = (np.arange(n) % 3 != 0).astype(np.uint8)
    m = f.density_map_nodes(n, h, 1000.0, True, mask=mask)
    np.testing.assert_array_equal(m[mask == 1], want[mask == 1])
    assert (m[mask == 0] == DBL_MAX).all()
    f.build_cell_major()
    np.testing.assert_array_equal(f.density_map_nodes(n, h, 1000.0, True), want)
    f.drop_cell_major()
    f.build_tile_major()     # explicit tile-major copy (the plain launches above built one per launch)
    np.testing.assert_array_equal(f.density_map_nodes(n, h, 1000.0, True), want)
    np.testing.assert_array_equal(f.density_map_nodes(n, h, 1000.0, True, 777, 2500), want[777:2500])
    f.drop_tile_major()
    T.force(monkeypatch, k3_tiles=0)   # and without any copy
    np.testing.assert_array_equal(f.density_map_nodes(n, h, 1000.0, True), want)
    T.force(monkeypatch, k3_tiles=None, k3_blocked=0)   # bricks in row-major instead of blocked order
    np.testing.assert_array_equal(f.density_map_nodes(n, h, 1000.0, True), want)
    np.testing.assert_array_equal(f.density_map_nodes(n, h, 1000.0, True, 777, 2500), want[777:2500])
    T.force(monkeypatch, k3_blocked=None)
    if key == "torus_density_h015":
        np.testing.assert_array_equal(f.density_map_nodes(n, h, 1000.0, False), golden["torus_density_h015_nopred"])


@pytest.mark.parametrize("tag,h", [("h012", 0.12), ("h045", 0.45)])
def test_density_map_on_a_deeper_grid_vs_reference_golden(dg, tag, h):
    """tests/golden/density_box.npz: torus 12 x 11 x 9, a small and a large support radius (up to
    four cells per h), made by the unmodified reference; ranges and masks included."""
    d = np.load(os.path.join(T.GOLDEN, "density_box.npz"))
    want = d["density_" + tag]
    grid = dg.grid_desc(d["domain"][:3], d["domain"][3:], d["res"])
    f = dg.Field(grid, d["sdf"])
    n = dg.n_nodes(grid)
    got = f.density_map_nodes(n, h, 1000.0, True)
    assert T.rel_err(got, want) <= 1e-10
    np.testing.assert_array_equal(got, want)
    np.testing.assert_array_equal(f.density_map_nodes(n, h, 1000.0, True, 1234, 4321), want[1234:4321])
    mask = (np.arange(n) % 5 != 0).astype(np.uint8)
    m = f.density_map_nodes(n, h, 1000.0, True, mask=mask)
    np.testing.assert_array_equal(m[mask == 1], want[mask == 1])
    assert (m[mask == 0] == DBL_MAX).all()


def test_density_map_without_predicate_on_a_bigger_grid(dg):
    """Icosphere SDF 40 x 36 x 33 (made by K1), h = 0.2 and no band predicate (every node near the
    surface integrates, domain boundaries included): == the oracle on a few hundred nodes, and the
    cell-major layout gives the same bits."""
    V, F = T.icosphere(8)
    dom = T.oracle_default_domain(V)
    res = [40, 36, 33]
    grid = dg.grid_desc(dom[:3], dom[3:], res)
    sdf = dg.Mesh(V, F).sample_nodes(grid)
    f = dg.Field(grid, sdf)
    n = dg.n_nodes(grid)
    got = f.density_map_nodes(n, 0.2, 1000.0, False)
    assert ((got != 0.0) & (got != DBL_MAX)).sum() > 50000
    for b in (0, n // 2, n - 150):
        np.testing.assert_array_equal(got[b:b + 150], T.oracle_density_map(dom, res, sdf, 0.2, 1000.0, False, b, b + 150))
    f.build_cell_major()
    np.testing.assert_array_equal(f.density_map_nodes(n, 0.2, 1000.0, False), got)


def test_point_lane_kernel_equals_the_brick_kernel_and_the_emulator(dg, monkeypatch):
    """Whole-lattice launches over an unreduced field take k_density_cells (a lane owns a lattice point with its seven
    nodes, waves of 16 x 2 x 2 points on the x-major copy of the Y / Z classes); DG_FORCE=k3_cells=0 is the brick kernel
    k_density_bricks (one node per lane; what reduced fields and short node ranges take), with and without its tile-major copy.
    Same bits from all of them, from other block shapes, on resolutions that are no multiples of the lane shape, with a node mask, and
    on fields spoilt with "no value" (answered by the copy's one bit per cell), NaN and Inf coefficients (no skipping of
    zero-weight points) -- checked against the host emulation of the product's arithmetic."""
    import emu
    V, F = T.icosphere(8)
    dom = T.oracle_default_domain(V)
    rng = np.random.default_rng(11)
    for res, h in (([21, 7, 10], 0.3), ([5, 35, 3], 0.25), ([40, 36, 33], 0.2)):
        grid = dg.grid_desc(dom[:3], dom[3:], res)
        sdf = dg.Mesh(V, F).sample_nodes(grid)
        n = dg.n_nodes(grid)
        spoilt = sdf.copy()
        spoilt[rng.integers(0, n, size=max(3, n // 3000))] = DBL_MAX
        worse = spoilt.copy()
        worse[rng.integers(0, n, size=3)] = np.nan
        worse[rng.integers(0, n, size=3)] = np.inf
        mask = (np.arange(n) % 7 != 0).astype(np.uint8)
        for name, coeffs in (("clean", sdf), ("no value", spoilt), ("nan / inf", worse)):
            f = dg.Field(grid, coeffs)
            got = {}
            for tag, env in (("cells", {}), ("cells, other blocks", {"k3_rb0": 3, "k3_rb1": 2, "k3_rb2": 5}),
                             ("cells, row-major", {"k3_blocked": 0}),
                             ("bricks", {"k3_cells": 0}), ("bricks, no copy", {"k3_cells": 0, "k3_tiles": 0})):
                T.force(monkeypatch, **env)
                got[tag] = (f.density_map_nodes(n, h, 1000.0, True), f.density_map_nodes(n, h, 1000.0, False, mask=mask))
                T.force(monkeypatch, **{k_: None for k_ in env})
            for tag in got:
                np.testing.assert_array_equal(got[tag][0], got["bricks"][0], err_msg="%s %s %s" % (res, name, tag))
                np.testing.assert_array_equal(got[tag][1], got["bricks"][1], err_msg="%s %s %s (mask)" % (res, name, tag))
            assert (got["cells"][1][mask == 0] == DBL_MAX).all()
            # node ranges: an eighth of the lattice or more stays with the point kernel (lanes outside the range idle), less goes to the brick kernel
            for b, e in ((n // 3, n - 5), (n // 2, n // 2 + n // 7), (17, 17 + n // 20)):
                np.testing.assert_array_equal(f.density_map_nodes(n, h, 1000.0, True, b, e), got["bricks"][0][b:e], err_msg="%s %s [%d, %d)" % (res, name, b, e))
                np.testing.assert_array_equal(f.density_map_nodes(n, h, 1000.0, False, b, e, mask=mask[b:e]), got["bricks"][1][b:e])
            if res[0] != 40:   # (the emulator walks every quadrature point of every node on the host)
                want = emu.density_map(dom, res, coeffs, h, 1000.0, band=True)
                np.testing.assert_array_equal(got["cells"][0], want, err_msg="%s %s" % (res, name))
            f.close()


def test_generate_density_map_cli(tmp_path):
    """GenerateDensityMap on the golden SDF files == the files the reference's tool flow writes
    (addFunction with predicate, both reduceField calls, save), byte for byte."""
    exe = os.path.join(BUILD, "GenerateDensityMap")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.dirname(BUILD)])
    out = str(tmp_path / "a.cdm")
    subprocess.check_call([exe, "-s", "0.1", "-r", "1000", "-o", out, os.path.join(T.GOLDEN, "torus_16_16_6.cdf")],
                          stdout=subprocess.DEVNULL)
    assert open(out, "rb").read() == open(os.path.join(T.GOLDEN, "torus_16_16_6_density_reduced.cdm"), "rb").read()
    out2 = str(tmp_path / "b.cdm")
    subprocess.check_call([exe, "--smoothing_length=0.15", "--no-reduction", "--output", out2,
                           os.path.join(T.GOLDEN, "torus_9_14_6.cdf")], stdout=subprocess.DEVNULL)
    got = T.read_cdf(out2)
    want = T.read_cdf(os.path.join(T.GOLDEN, "torus_9_14_6_density.cdm"))
    # the golden was produced WITH the predicate (which rejects nothing on this coarse lattice)
    np.testing.assert_array_equal(got["nodes"][1], want["nodes"][1])
    assert open(out2, "rb").read() == open(os.path.join(T.GOLDEN, "torus_9_14_6_density.cdm"), "rb").read()


def test_density_map_bigger_lattice_vs_oracle(dg):
    """Icosphere SDF at 40^3 (GPU-generated), density map with the tool's defaults (h = 0.1):
    random node sample against the oracle; timing printed."""
    V, F = T.icosphere(12)
    dom = T.oracle_default_domain(V)
    res = [40, 40, 40]
    grid = dg.grid_desc(dom[:3], dom[3:], res)
    sdf = dg.Mesh(V, F).sample_nodes(grid)
    f = dg.Field(grid, sdf)
    n = dg.n_nodes(grid)
    t0 = time.time()
    got = f.density_map_nodes(n, 0.1, 1000.0, True)
    dt = time.time() - t0
    active = int(((got != DBL_MAX) & (got != 0)).sum())
    print("density map 40^3: %d nodes, %d integrated, K3 %.1f ms (%.2f G interpolations/s), wall %.2f s"
          % (n, active, dg.last_kernel_ms(), active * 4097 / dg.last_kernel_ms() / 1e6, dt))
    rng = np.random.default_rng(3)
    for b in rng.integers(0, n - 64, size=12):
        want = T.oracle_density_map(dom, res, sdf, 0.1, 1000.0, True, int(b), int(b) + 64)
        np.testing.assert_array_equal(got[b:b + 64], want)
    assert active > 20000


def test_point_lane_kernel_beyond_two_gigabytes_of_offsets(dg, monkeypatch):
    """k_density_cells addresses a cell by 32-bit byte offsets on scalar row bases.  At 512^3 the X class alone is 2.16 GB and the
    x-major copy of the Y / Z classes 2 x 2.15 GB: offsets beyond 2^31 must be taken as UNSIGNED by the load.  Node ranges at
    the far end of every class (where the offsets are largest), point-lane kernel == brick kernel, bit for bit."""
    import torch
    V, F = T.icosphere(24)
    dom = T.oracle_default_domain(V)
    res = [512, 512, 512]
    grid = dg.grid_desc(dom[:3], dom[3:], res)
    n = dg.n_nodes(grid)
    s = torch.cuda.current_stream().cuda_stream
    sdf = torch.empty(n, dtype=torch.float64, device="cuda")
    dg.Mesh(V, F).sample_nodes_device(grid, 0, n, sdf.data_ptr(), stream=s)
    fld = dg.Field(grid, d_coeffs=sdf.data_ptr(), n_coeffs=n)
    nv = 513 ** 3
    ne2 = 2 * 512 * 513 * 513
    m = n // 8 + 1000                      # (an eighth of the lattice or more: the whole-lattice kernels)
    outs = {}
    for begin in (nv + ne2 - m, nv + 2 * ne2 - m, n - m):      # the ends of the X, Y and Z classes
        for tag, cells in (("cells", "1"), ("bricks", "0")):
            T.force(monkeypatch, k3_cells=cells)
            out = torch.full((m,), -1.0, dtype=torch.float64, device="cuda")
            fld.density_map_nodes_device(0.05, 1000.0, True, begin, begin + m, out.data_ptr(), stream=s)
            torch.cuda.synchronize()
            outs[tag] = out
        T.force(monkeypatch, k3_cells=None)
        assert torch.equal(outs["cells"], outs["bricks"]), begin
        got = outs["cells"]
        assert int(((got != DBL_MAX) & (got != 0.0)).sum().item()) > 100000      # (the band is there: real quadratures ran)
    fld.close()
