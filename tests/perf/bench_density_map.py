#!/usr/bin/env python3
"""Secondary benchmark: K3, the SPH boundary density map (BASELINE.json configs[4], "GenerateDensityMap
cubic-spline kernel on a precomputed SDF"): icosphere nu=71 SDF on an N^3 grid (default 128^3, generated on
the GPU by K1), support radius h = 0.1, rho0 = 1000, band predicate on.  One JSON line: lattice nodes/s,
G interpolations/s (4097 per integrated node) and the CPU restatement on a bounded node sample."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=128)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--cpu-nodes", type=int, default=4096)
    args = ap.parse_args()
    import torch
    import dgtest as T
    import discregrid_amd as dg

    dg.load_library()
    V, F = T.icosphere(71)
    dom = T.oracle_default_domain(V)
    res = [args.res] * 3
    grid = dg.grid_desc(dom[:3], dom[3:], res)
    n = dg.n_nodes(grid)
    s = torch.cuda.current_stream().cuda_stream
    sdf = torch.empty(n, dtype=torch.float64, device="cuda")
    dg.Mesh(V, F).sample_nodes_device(grid, 0, n, sdf.data_ptr(), stream=s)
    torch.cuda.synchronize()
    field = dg.Field(grid, d_coeffs=sdf.data_ptr(), n_coeffs=n)
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    h, rho0 = 0.1, 1000.0
    results = {}
    for layout in ("node-order", "tile-major", "cell-major"):
        if layout == "tile-major":
            field.build_tile_major(s)
        if layout == "cell-major":
            field.drop_tile_major()
            field.build_cell_major(s)
        field.density_map_nodes_device(h, rho0, True, 0, n, out.data_ptr(), stream=s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            field.density_map_nodes_device(h, rho0, True, 0, n, out.data_ptr(), stream=s)
        e1.record()
        torch.cuda.synchronize()
        results[layout] = e0.elapsed_time(e1) / args.steps
    got = out.cpu().numpy()
    big = np.finfo(np.float64).max
    integrated = int(((got != big) & (got != 0.0)).sum())
    ms = min(results.values())
    line = {"metric": "density-map lattice nodes/s (K3)", "value": n / ms / 1e3, "unit": "Mnodes/s", "ms": ms,
            "ms_by_layout": results, "grid": res, "nodes": n, "integrated_nodes": integrated,
            "rejected_by_predicate": int((got == big).sum()),
            "G_interpolations_per_s": integrated * 4097 / ms / 1e6, "h": h, "rho0": rho0, "dtype": "f64"}
    if args.cpu_nodes > 0:
        coeffs = sdf.cpu().numpy()
        idx = np.flatnonzero((got != big) & (got != 0.0))
        b = int(idx[len(idx) // 2])
        e = min(n, b + args.cpu_nodes)
        want = T.oracle_density_map(dom, res, coeffs, h, rho0, True, b, e)
        secs = T.oracle_density_map.last_seconds
        k = int(((want != big) & (want != 0.0)).sum())
        line["bit_exact_vs_oracle_on_sample"] = bool(np.array_equal(want, got[b:e]))
        line["cpu_baseline"] = {"value": k * 4097 / secs / 1e9, "unit": "G interpolations/s", "cores": os.cpu_count(),
                                "kind": "port", "sample": "%d consecutive lattice nodes (%d integrated), %.1f s" % (e - b, k, secs)}
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
