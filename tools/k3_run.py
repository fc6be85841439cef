#!/usr/bin/env python3
"""K3 workload for profiling / A-B runs: the density map on the icosphere SDF at N^3 (default 128), `steps` launches.
    python tools/k3_run.py [--res 128] [--steps 2] [--check]     prints ms per launch (HIP events) as JSON
--check compares every block of 2^20 results with tests/golden/lattice_digests.npz (128 and 256 only)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", type=int, default=128)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--sweep", default="", help="';'-separated settings, each 'NAME=v,NAME=v': one measurement per setting")
    a = ap.parse_args()
    import torch
    import dgtest as T
    import discregrid_amd as dg
    dg.load_library()
    V, F = T.icosphere(71)
    dom = dg.default_domain(V)
    grid = dg.grid_desc(dom[:3], dom[3:], [a.res] * 3)
    n = dg.n_nodes(grid)
    s = torch.cuda.current_stream().cuda_stream
    sdf = torch.empty(n, dtype=torch.float64, device="cuda")
    dg.Mesh(V, F).sample_nodes_device(grid, 0, n, sdf.data_ptr(), stream=s)
    fld = dg.Field(grid, d_coeffs=sdf.data_ptr(), n_coeffs=n)
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    fld.density_map_nodes_device(0.1, 1000.0, True, 0, min(n, 1 << 18), out.data_ptr(), stream=s)
    torch.cuda.synchronize()
    gold = np.load(os.path.join(ROOT, "tests", "golden", "lattice_digests.npz"))
    key = "density%d_digest" % a.res
    for setting in (a.sweep.split(";") if a.sweep else [""]):
        names = []
        for kv in filter(None, setting.split(",")):      # keys of DG_FORCE (k3_rb1=6, k3_cells=0, ...: discregrid_amd/csrc/dg_force.h)
            name, v = kv.split("=")
            names.append(name)
        os.environ["DG_FORCE"] = ";".join(filter(None, setting.split(",")))
        out.fill_(-1.0)
        ms = []
        for _ in range(a.steps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fld.density_map_nodes_device(0.1, 1000.0, True, 0, n, out.data_ptr(), stream=s)
            e1.record()
            torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        res = {"res": a.res, "setting": setting, "ms": [round(m, 2) for m in ms], "best_ms": min(ms)}
        if a.check:
            if key in gold:
                got = T.block_digests(out.cpu().numpy())
                res["mismatching_blocks"] = int((got != gold[key]).any(axis=1).sum())
            else:
                res["mismatching_blocks"] = None
        print(json.dumps(res), flush=True)
        os.environ.pop("DG_FORCE", None)

if __name__ == "__main__":
    main()
