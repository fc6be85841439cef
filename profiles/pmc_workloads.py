#!/usr/bin/env python3
"""Workloads profiles/collect.sh profiles besides the default bench: K2 (batched interpolate), K3 (density
map) and U (unpack) at BASELINE configs[4] / configs[2] sizes on the 256^3 icosphere field.
    python profiles/pmc_workloads.py k2|k2r|k2b|k3|u
Measurement tooling (uses the test helpers for the mesh and the config-5 query generator)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    what = sys.argv[1]
    import torch
    import dgtest as T
    import discregrid_amd as dg
    dg.load_library()
    dg.set_device(0)
    V, F = T.icosphere(71)
    dom = dg.default_domain(V)
    res = [256, 256, 256]
    grid = dg.grid_desc(dom[:3], dom[3:], res)
    n = dg.n_nodes(grid)
    mesh = dg.Mesh(V, F)
    s = torch.cuda.current_stream().cuda_stream
    field = torch.empty(n, dtype=torch.float64, device="cuda")
    mesh.sample_nodes_device(grid, 0, n, field.data_ptr(), stream=s)
    torch.cuda.synchronize()
    if what == "k2":
        fld = dg.Field(grid, d_coeffs=field.data_ptr(), n_coeffs=n)
        nq = 10_000_000
        P = torch.from_numpy(T.uniform_points(1234, nq, dom[:3], dom[3:])).cuda()
        phi = torch.empty(nq, dtype=torch.float64, device="cuda")
        grad = torch.empty(3 * nq, dtype=torch.float64, device="cuda")
        for rep in range(3):       # unordered input: probe + keys + radix sort + binned kernel
            fld.interpolate_device(P.data_ptr(), nq, phi.data_ptr(), stream=s)
        for rep in range(3):
            fld.interpolate_device(P.data_ptr(), nq, phi.data_ptr(), grad.data_ptr(), stream=s)
        torch.cuda.synchronize()
        # the same queries sorted by cell: the plain kernel (no sort launched once the probe says "ordered")
        h = torch.tensor((dom[3:] - dom[:3]) / 256.0, device="cuda")
        cell = ((P - torch.tensor(dom[:3], device="cuda")) / h).floor().clamp(0, 255).long()
        key = (cell[:, 2] * 256 + cell[:, 1]) * 256 + cell[:, 0]
        Ps = P[torch.argsort(key)].contiguous()
        for rep in range(4):
            fld.interpolate_device(Ps.data_ptr(), nq, phi.data_ptr(), stream=s)
        torch.cuda.synchronize()
    elif what == "k2r":
        # K2 through the cooperative row kernel on the cell-major copy: unordered queries, no binning
        fld = dg.Field(grid, d_coeffs=field.data_ptr(), n_coeffs=n)
        fld.build_cell_major(s)
        nq = 10_000_000
        P = torch.from_numpy(T.uniform_points(1234, nq, dom[:3], dom[3:])).cuda()
        phi = torch.empty(nq, dtype=torch.float64, device="cuda")
        grad = torch.empty(3 * nq, dtype=torch.float64, device="cuda")
        for rep in range(3):
            fld.interpolate_device(P.data_ptr(), nq, phi.data_ptr(), stream=s)
        for rep in range(3):
            fld.interpolate_device(P.data_ptr(), nq, phi.data_ptr(), grad.data_ptr(), stream=s)
        torch.cuda.synchronize()
    elif what == "k2b":
        # K2 through the BAND-LIMITED cell-major copy: the SPH-like shell |phi| < 2h (every query has a row) and uniform queries
        # (four in ten gather from the field), unordered, no binning
        import numpy as np
        fld = dg.Field(grid, d_coeffs=field.data_ptr(), n_coeffs=n)
        diag = float(np.linalg.norm((dom[3:] - dom[:3]) / 256.0))
        nq = 10_000_000
        C = torch.from_numpy(T.uniform_points(4321, 26_000_000, dom[:3], dom[3:])).cuda()
        phic = torch.empty(len(C), dtype=torch.float64, device="cuda")
        fld.interpolate_device(C.data_ptr(), len(C), phic.data_ptr(), stream=s)     # (plain path: the band copy does not exist yet)
        S = C[(phic.abs() < 0.2)][:nq].contiguous()
        del C, phic
        fld.build_cell_major_band(-(0.2 + diag), 0.2 + diag, s)    # from here on k_interpolate_band sees the shell batches only
        phi = torch.empty(nq, dtype=torch.float64, device="cuda")
        grad = torch.empty(3 * nq, dtype=torch.float64, device="cuda")
        for rep in range(3):
            fld.interpolate_device(S.data_ptr(), nq, phi.data_ptr(), stream=s)
        for rep in range(3):
            fld.interpolate_device(S.data_ptr(), nq, phi.data_ptr(), grad.data_ptr(), stream=s)
        torch.cuda.synchronize()
    elif what == "k3":
        fld = dg.Field(grid, d_coeffs=field.data_ptr(), n_coeffs=n)
        dens = torch.empty(n, dtype=torch.float64, device="cuda")
        for rep in range(2):
            fld.density_map_nodes_device(0.1, 1000.0, True, 0, n, dens.data_ptr(), stream=s)
        torch.cuda.synchronize()
    elif what == "u":
        # the unpack kernels of the multi-GPU path on one GPU: 8 shards sampled in turn, unpacked whole and by ranges
        nr = 8
        cnt, stride = dg.shard_layout(grid, 0, nr)
        gathered = torch.zeros(nr * stride, dtype=torch.float64, device="cuda")
        for r in range(nr):
            mesh.sample_shard_device(grid, r, nr, gathered[r * stride:].data_ptr(), stream=s)
        out = torch.empty(n, dtype=torch.float64, device="cuda")
        for rep in range(3):
            dg.unpack_shards_device(grid, nr, gathered.data_ptr(), stride, out.data_ptr(), stream=s)
        for rep in range(3):
            for r0 in range(0, nr, 2):
                dg.unpack_shard_range_device(grid, nr, gathered.data_ptr(), stride, r0, r0 + 2, out.data_ptr(), stream=s)
        torch.cuda.synchronize()
        assert torch.equal(out, field)
    else:
        raise SystemExit("k2 | k2r | k2b | k3 | u")


if __name__ == "__main__":
    main()
