#!/usr/bin/env python3
"""Randomised parity campaign (run on the GPU box): random meshes x random lattices x random point
sets, K1 / K1p / K2 / K3 through the C ABI against the CPU oracle, bit for bit (unsigned distance for
meshes that are not closed).  usage: fuzz_parity.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def random_mesh(rng):
    import dgtest as T
    kind = rng.integers(0, 6)
    if kind == 0:
        V, F = T.icosphere(int(rng.integers(1, 14)))
        V = V * rng.uniform(0.2, 3.0, size=3) + rng.normal(scale=10.0, size=3)          # ellipsoid somewhere
        return "ellipsoid", V, F, True
    if kind == 1:
        V, F = T.torus(int(rng.integers(5, 60)), int(rng.integers(4, 30)), 1.0, float(rng.uniform(0.1, 0.6)))
        return "torus", V, F, True
    if kind == 2:
        V, F = T.icosphere(int(rng.integers(2, 10)))
        V = V * (1.0 + 0.3 * rng.normal(size=(len(V), 1)).clip(-2, 2))                   # spiky star-shaped blob
        return "blob", V, F, True
    if kind == 3:
        n = int(rng.integers(1, 300))
        V = rng.uniform(-1, 1, size=(3 * n, 3)) * rng.uniform(0.01, 2.0)
        return "soup", V, np.arange(3 * n, dtype=np.uint32).reshape(n, 3), False
    if kind == 4:
        V, F = T.box_mesh()
        V = V * rng.uniform(1e-3, 1e3, size=3)
        return "box", V, F, True
    V, F = T.bunny_mesh()
    keep = rng.random(len(F)) < rng.uniform(0.05, 1.0)                                    # bunny with holes
    return "holey bunny", V, F[keep], bool(keep.all())


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    import dgtest as T
    import discregrid_amd as dg
    dg.load_library()
    rng = np.random.default_rng(seed)
    t0 = time.time()
    rounds = nodes = points = 0
    k3_rounds = [0]
    while time.time() - t0 < budget:
        name, V, F, closed = random_mesh(rng)
        om, m = T.OracleMesh(V, F), dg.Mesh(V, F)
        lo, hi = V.min(axis=0), V.max(axis=0)
        ext = np.maximum(hi - lo, 1e-9 * max(np.abs(hi).max(), 1e-300))
        pad = rng.uniform(-0.3, 1.0, size=3) * ext                                        # the domain may cut the mesh
        dom = np.concatenate([lo - np.abs(pad) * (pad > 0) + np.abs(pad) * (pad < 0) * 0.2, hi + pad.clip(0.01 * ext.max())])
        res = [int(x) for x in rng.integers(1, 28, size=3)]
        grid = dg.grid_desc(dom[:3], dom[3:], res)
        got, want = m.sample_nodes(grid), om.sample_nodes(dom, res)
        if closed:
            onsurf = np.abs(want) < 1e-9 * ext.max()
            ok = np.array_equal(got[~onsurf], want[~onsurf]) and np.array_equal(np.abs(got), np.abs(want))
        else:
            ok = np.array_equal(np.abs(got), np.abs(want))
        P = rng.uniform(lo - ext, hi + ext, size=(int(rng.integers(1, 9000)), 3))
        a, b = m.signed_distance(P), om.signed_distance(P)
        ok = ok and np.array_equal(np.abs(a), np.abs(b))
        f = dg.Field(grid, got)
        Q = rng.uniform(dom[:3] - 0.05 * ext, dom[3:] + 0.05 * ext, size=(5000, 3))
        phi, grad = f.interpolate(Q, grad=True)
        wphi, wgrad = T.oracle_interpolate(dom, res, want, Q, grad=True)
        inside = wphi != np.finfo(np.float64).max
        same_field = np.array_equal(got, want)
        if same_field:
            ok = ok and np.array_equal(phi, wphi) and np.array_equal(grad[inside], wgrad[inside])
        # K2 through the staged gather of the plain layout (round 6: sort by tile of 8^3 cells, tile image in LDS), forced for this small
        # batch, with and without the XCD-aware item order; and through the radix-sorted per-lane gather of rounds 1-5: the same bits
        if rounds % 2 == 0:
            for force in (dict(k2_binning=2, k2_tiles=2), dict(k2_binning=2, k2_tiles=2, k2_tile_chunk=int(rng.integers(0, 5))), dict(k2_binning=2, k2_tiles=0)):
                os.environ["DG_FORCE"] = T.force_string(os.environ.get("DG_FORCE"), **force)
                phi_t, grad_t = f.interpolate(Q, grad=True)
                ok = ok and np.array_equal(phi_t, phi) and np.array_equal(grad_t, grad) and np.array_equal(f.interpolate(Q), phi)
                os.environ["DG_FORCE"] = T.force_string(os.environ.get("DG_FORCE"), **{k: None for k in force})
        # K1 through the FILTERED kernel whatever the mesh's size, its epilogue's pool capped at random (round 6): the same lattice
        if rounds % 5 == 0:
            os.environ["DG_FORCE"] = T.force_string(os.environ.get("DG_FORCE"), k1_fast=1, pool_cap=int(rng.integers(0, 400)))
            ok = ok and np.array_equal(m.sample_nodes(grid), got)
            os.environ["DG_FORCE"] = T.force_string(os.environ.get("DG_FORCE"), k1_fast=None, pool_cap=None)
        # K2 through a band-limited cell-major copy with a random band (round 4): the same bits
        if rounds % 3 == 0:
            fin = got[np.isfinite(got) & (got != np.finfo(np.float64).max)]
            if len(fin):
                lo_b = float(rng.uniform(fin.min() - 0.1, fin.max() + 0.1))
                hi_b = lo_b + float(rng.uniform(0.0, 1.0)) * float(fin.max() - fin.min() + 1e-9)
                f.build_cell_major_band(lo_b, hi_b)
                phi_b, grad_b = f.interpolate(Q, grad=True)
                ok = ok and np.array_equal(phi_b, phi) and np.array_equal(grad_b, grad)
                f.drop_cell_major()
        # K3 on a slice of the lattice, now and then on a field spoilt with NaN / Inf / huge / DBL_MAX values
        # (those must take the evaluate-every-point path and still match the oracle, NaNs included)
        if rounds % 4 == 0 and len(got) > 64:
            sdf = want.copy()
            spoil = rng.integers(0, 4)
            if spoil == 1:
                sdf[rng.integers(0, len(sdf), size=3)] = [np.nan, np.inf, -1.0e300]
            elif spoil == 2:
                sdf[rng.integers(0, len(sdf), size=5)] = np.finfo(np.float64).max
            h = float(rng.uniform(0.05, 0.6)) * float(ext.max())
            band = bool(rng.integers(0, 2))
            b = int(rng.integers(0, len(sdf) - 60))
            e = b + int(rng.integers(1, 60))
            fk = dg.Field(grid, sdf)
            k3 = fk.density_map_nodes(len(sdf), h, 1000.0, band, b, e)
            w3 = T.oracle_density_map(dom, res, sdf, h, 1000.0, band, b, e)
            ok = ok and np.array_equal(k3, w3, equal_nan=True)
            # the whole lattice goes through the point-lane kernel (k_density_cells), a small slice through the brick kernel:
            # the slice of the one == the oracle, and the whole == the brick kernel's whole
            full = fk.density_map_nodes(len(sdf), h, 1000.0, band)
            os.environ["DG_FORCE"] = T.force_string(os.environ.get("DG_FORCE"), k3_cells=0)
            full_bricks = fk.density_map_nodes(len(sdf), h, 1000.0, band)
            os.environ["DG_FORCE"] = T.force_string(os.environ.get("DG_FORCE"), k3_cells=None)
            ok = ok and np.array_equal(full[b:e], w3, equal_nan=True) and np.array_equal(full, full_bricks, equal_nan=True)
            k3_rounds[0] += 1
        rounds += 1
        nodes += len(got)
        points += len(P)
        if not ok:
            print("MISMATCH in round %d (seed %d): %s, %d triangles, res %s" % (rounds, seed, name, len(F), res))
            sys.exit(1)
    print("fuzz ok: %d rounds (%d with a density-map slice), %d lattice nodes, %d points, seed %d, %.0f s" % (
        rounds, k3_rounds[0], nodes, points, seed, time.time() - t0))


if __name__ == "__main__":
    main()
