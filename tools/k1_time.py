#!/usr/bin/env python3
"""K1 kernel time of ONE library build on the judged lattices (icosphere nu=71 256^3, bunny 256^3, dragon 256^3): HIP events
around dg_sdf_sample_nodes_device, mean of --reps launches after a warm-up.  DG_LIB selects the library (same-box A/B of
variant builds: tools/gpu_k1_libs.sh).  Prints one JSON line."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--meshes", default="ico256,bunny256,dragon256")
    a = ap.parse_args()
    import numpy as np
    import torch
    import dgtest as T
    import discregrid_amd as dg
    dg.load_library()
    dg.set_device(0)
    s = torch.cuda.current_stream().cuda_stream
    out = {"lib": os.environ.get("DG_LIB", "in-tree"), "force": os.environ.get("DG_FORCE", "")}
    for name in a.meshes.split(","):
        make = {"ico": lambda: T.icosphere(71), "bunny": T.bunny_mesh, "dragon": T.dragon_mesh}[name.rstrip("0123456789")]
        r = int(name[len(name.rstrip("0123456789")):])
        V, F = make()
        dom = dg.default_domain(V)
        grid = dg.grid_desc(dom[:3], dom[3:], [r] * 3)
        n = dg.n_nodes(grid)
        mesh = dg.Mesh(V, F)
        buf = torch.empty(n, dtype=torch.float64, device="cuda")
        for _ in range(3):
            mesh.sample_nodes_device(grid, 0, n, buf.data_ptr(), stream=s)
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.reps)]
        for e0, e1 in ev:
            e0.record()
            mesh.sample_nodes_device(grid, 0, n, buf.data_ptr(), stream=s)
            e1.record()
        torch.cuda.synchronize()
        ms = [e0.elapsed_time(e1) for e0, e1 in ev]
        out[name] = {"ms": round(float(np.mean(ms)), 4), "min_ms": round(float(np.min(ms)), 4), "checksum": float(buf[::100003].sum().item())}
        del buf, mesh
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
