#!/usr/bin/env python3
"""Same-box A/B of K2 on the PLAIN layout (attached device array, no copy of the field): the radix-sorted per-lane gather of
rounds 1-5 (DG_FORCE=k2_tiles=0) against round 6's counting sort by tile + LDS-staged gather, with its knobs
(k2_tile_chunk: consecutive work items per XCD; k2_stage_min: fewest queries of an item that is staged).  10 M queries, uniform
(std::mt19937_64 seed 1234) and the |phi| < 2h shell, value and value + gradient, interleaved rounds; every variant's results are
compared with the first's, bit for bit.  One JSON line per variant.

    python tests/perf/k2_plain_ab.py [--rounds 3] [--variants "k2_tiles=0|k2_tile_chunk=64|..."]
"""
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
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--queries", type=int, default=10_000_000)
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--variants", default="k2_tiles=0|k2_tiles=1|k2_tile_chunk=0|k2_tile_chunk=16|k2_tile_chunk=256|k2_stage_min=0|k2_stage_min=100")
    ap.add_argument("--once", action="store_true", help="one call per variant and case, no timing loop (profiling runs)")
    ap.add_argument("--ordered", action="store_true", help="ordered batches instead: the uniform queries in z-order and in row order of their cells")
    ap.add_argument("--band", action="store_true", help="build the band-limited cell-major copy (|phi| <= 2h + cell diagonal, h = 0.1) first: the variants then "
                                                        "compare routings of a field WITH that copy (k2_band_split=0|1, k2_band=0)")
    args = ap.parse_args()
    import torch
    import dgtest as T
    import discregrid_amd as dg

    dg.load_library()
    V, F = T.icosphere(71)
    dom = dg.default_domain(V)
    res = [args.res] * 3
    grid = dg.grid_desc(dom[:3], dom[3:], res)
    n = dg.n_nodes(grid)
    mesh = dg.Mesh(V, F)
    stream = torch.cuda.current_stream()
    s = stream.cuda_stream
    field = torch.empty(n, dtype=torch.float64, device="cuda")
    mesh.sample_nodes_device(grid, 0, n, field.data_ptr(), stream=s)
    torch.cuda.synchronize()
    fld = dg.Field(grid, d_coeffs=field.data_ptr(), n_coeffs=n)
    nq = args.queries
    P = torch.from_numpy(T.uniform_points(1234, nq, dom[:3], dom[3:])).cuda()
    C = torch.from_numpy(T.uniform_points(4321, int(2.6 * nq), dom[:3], dom[3:])).cuda()
    phic = torch.empty(len(C), dtype=torch.float64, device="cuda")
    fld.interpolate_device(C.data_ptr(), len(C), phic.data_ptr(), stream=s)
    S = C[(phic.abs() < 0.2)][:nq].contiguous()
    del C, phic
    phi = torch.empty(nq, dtype=torch.float64, device="cuda")
    grad = torch.empty(3 * nq, dtype=torch.float64, device="cuda")
    if args.band:
        diag = float(np.linalg.norm((dom[3:] - dom[:3]) / np.array(res, dtype=np.float64)))
        rows = fld.build_cell_major_band(-(0.2 + diag), 0.2 + diag, s)
        torch.cuda.synchronize()
        print(json.dumps({"band_rows": rows, "fraction_of_cells": rows / float(np.prod(res))}), flush=True)
    variants = [v for v in args.variants.split("|") if v]
    cases = [("uniform", P, False), ("uniform", P, True), ("shell", S, False), ("shell", S, True)]
    if args.ordered:
        # the same uniform queries as a caller with spatially sorted particles hands them over: in z-order of their cells (SPlisHSPlasH's sort) and in
        # row order of their cells
        h3 = torch.tensor((dom[3:] - dom[:3]) / np.array(res, dtype=np.float64), device="cuda")
        cell = ((P - torch.tensor(dom[:3], device="cuda")) / h3).floor().clamp(0, res[0] - 1).long()
        z = torch.zeros(nq, dtype=torch.long, device="cuda")
        for b in range(10):
            for d in range(3):
                z |= ((cell[:, d] >> b) & 1) << (3 * b + d)
        Pz = P[torch.argsort(z)].contiguous()
        Pr = P[torch.argsort((cell[:, 2] * res[1] + cell[:, 1]) * res[0] + cell[:, 0])].contiguous()
        del cell, z
        cases = [("zorder", Pz, False), ("zorder", Pz, True), ("rows", Pr, False), ("rows", Pr, True), ("uniform", P, False)]
    base = os.environ.get("DG_FORCE")
    times = {v: {"%s_%s" % (c[0], "grad" if c[2] else "value"): [] for c in cases} for v in variants}
    ref = {}
    equal = {v: True for v in variants}
    for rnd in range(1 if args.once else args.rounds):
        for v in variants:
            os.environ["DG_FORCE"] = ";".join(x for x in (base, v.replace(",", ";")) if x)
            for name, Q, g in cases:
                key = "%s_%s" % (name, "grad" if g else "value")
                fn = (lambda: fld.interpolate_device(Q.data_ptr(), nq, phi.data_ptr(), grad.data_ptr() if g else 0, stream=s))
                phi.fill_(float("nan"))
                fn()
                torch.cuda.synchronize()
                fn()                      # (routing probes: the verdict of the previous batch routes the next)
                torch.cuda.synchronize()
                if rnd == 0:
                    got = (phi.clone(), grad.clone() if g else None)
                    if key not in ref:
                        ref[key] = got
                    else:
                        equal[v] &= bool(torch.equal(got[0], ref[key][0])) and (not g or bool(torch.equal(got[1], ref[key][1])))
                if args.once:
                    continue
                fn()
                torch.cuda.synchronize()
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
                for a, b in ev:
                    a.record(stream)
                    fn()
                    b.record(stream)
                torch.cuda.synchronize()
                times[v][key].append(float(np.mean([a.elapsed_time(b) for a, b in ev])))
    for v in variants:
        rec = {"variant": v, "bit_equal_to_first_variant": equal[v]}
        for k, ts in times[v].items():
            if ts:
                rec[k + "_ms"] = [round(t, 4) for t in ts]
                rec[k + "_gq_s"] = round(nq / (min(ts) * 1e-3) / 1e9, 3)
        print(json.dumps(rec), flush=True)
    if base is None:
        os.environ.pop("DG_FORCE", None)
    else:
        os.environ["DG_FORCE"] = base
    fld.close()


if __name__ == "__main__":
    main()
