#!/usr/bin/env python3
"""Secondary benchmark (BASELINE.json configs[4]): batched interpolate() -- kernel K2 -- on the
256^3 icosphere SDF, 10 M query points, value-only and value+gradient, two distributions:
uniform in the domain (seed 1234) and an "SPH-like" shell |phi| < 2h, h = 0.1.  Prints one JSON
line per case (Mqueries/s, algorithmic GB/s = 288 or 312 B/query) and, with
--cpu-seconds > 0, the reference's OpenMP interpolate on a bounded sample of the same queries.

    python tests/perf/bench_interpolate.py [--queries 10000000] [--steps 5] [--cpu-seconds 5]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=10_000_000)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--cpu-seconds", type=float, default=5.0)
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
    mesh = dg.Mesh(V, F)
    s = torch.cuda.current_stream().cuda_stream
    field_t = torch.empty(n, dtype=torch.float64, device="cuda")
    mesh.sample_nodes_device(grid, 0, n, field_t.data_ptr(), stream=s)
    torch.cuda.synchronize()
    field = dg.Field(grid, d_coeffs=field_t.data_ptr(), n_coeffs=n)

    m = args.queries
    g = torch.Generator(device="cuda")
    g.manual_seed(1234)
    lo = torch.tensor(dom[:3], device="cuda")
    hi = torch.tensor(dom[3:], device="cuda")
    uni = lo + (hi - lo) * torch.rand((m, 3), dtype=torch.float64, device="cuda", generator=g)
    # shell |phi| < 2h around the unit sphere (h = 0.1): rejection-free construction
    dirs = torch.randn((m, 3), dtype=torch.float64, device="cuda", generator=g)
    dirs /= dirs.norm(dim=1, keepdim=True)
    rad = 1.0 + 0.2 * (2 * torch.rand((m, 1), dtype=torch.float64, device="cuda", generator=g) - 1)
    shell = (dirs * rad).clamp(lo, hi).contiguous()
    # the same uniform points sorted by grid cell: what a caller with spatially sorted particles sees
    cell = ((uni - lo) / (hi - lo) * args.res).long().clamp_(0, args.res - 1)
    key = (cell[:, 2] * args.res + cell[:, 1]) * args.res + cell[:, 0]
    uni_sorted = uni[torch.argsort(key)].contiguous()
    del cell, key
    phi = torch.empty(m, dtype=torch.float64, device="cuda")
    grad = torch.empty((m, 3), dtype=torch.float64, device="cuda")

    cases = []
    for layout in ("node-order gather", "tile-major", "cell-major"):
      if layout == "tile-major":
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        field.build_tile_major(s)
        torch.cuda.synchronize()
        build_ms = (time.perf_counter() - t0) * 1e3
      if layout == "cell-major":
        field.drop_tile_major()
        t0 = time.perf_counter()
        field.build_cell_major(s)
        torch.cuda.synchronize()
        build_ms = (time.perf_counter() - t0) * 1e3
      for dist_name, pts in (("uniform", uni), ("shell |phi|<2h", shell), ("uniform, cell-sorted", uni_sorted)):
        for with_grad in (False, True):
            for _ in range(2):
                field.interpolate_device(pts.data_ptr(), m, phi.data_ptr(), grad.data_ptr() if with_grad else 0, s)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                field.interpolate_device(pts.data_ptr(), m, phi.data_ptr(), grad.data_ptr() if with_grad else 0, s)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.steps
            bpq = 312 if with_grad else 288
            out = {"metric": "Mqueries/s batched interpolate", "value": m / ms / 1e3, "unit": "Mqueries/s",
                   "queries": m, "distribution": dist_name, "gradient": with_grad, "layout": layout, "ms": ms,
                   "dtype": "f64",
                   "field": "icosphere nu=71 SDF, %d^3 grid (%d coefficients, %.2f GB)" % (args.res, n, n * 8 / 1e9),
                   "roofline": {"bound": "hbm", "achieved": bpq * m / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                                "frac": bpq * m / (ms * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes_per_query": bpq}}
            if layout == "cell-major":
                out["cell_major_build_ms"] = build_ms
                out["cell_major_bytes"] = 256 * int(np.prod(res))
            if layout == "tile-major":
                out["tile_major_build_ms"] = build_ms
                out["tile_major_bytes"] = 736 * 8 * int(np.prod([(r + 3) // 4 for r in res]))
            cases.append((out, pts, with_grad))
    # parity spot check + CPU baseline on a bounded sample
    coeffs = None
    for out, pts, with_grad in cases:
        if args.cpu_seconds > 0:
            if coeffs is None:
                coeffs = field_t.cpu().numpy()
            k = min(m, 2_000_000)
            P = pts[:k].cpu().numpy()
            t0 = time.perf_counter()
            if with_grad:
                want, wg = T.oracle_interpolate(dom, res, coeffs, P, grad=True)
            else:
                want = T.oracle_interpolate(dom, res, coeffs, P)
            secs = T.oracle_interpolate.last_seconds
            field.interpolate_device(pts.data_ptr(), m, phi.data_ptr(), grad.data_ptr() if with_grad else 0, s)
            torch.cuda.synchronize()
            got = phi[:k].cpu().numpy()
            out["bit_exact_vs_oracle_on_sample"] = bool(np.array_equal(got, want))
            if with_grad:
                out["bit_exact_vs_oracle_on_sample"] &= bool(np.array_equal(grad[:k].cpu().numpy(), wg))
            out["cpu_baseline"] = {"value": k / secs / 1e6, "unit": "Mqueries/s", "cores": os.cpu_count(), "kind": "port",
                                   "sample": "%d of the same queries, OpenMP schedule(static)" % k}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
