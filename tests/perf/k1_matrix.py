#!/usr/bin/env python3
"""K1 over a matrix of meshes and lattice sizes (robustness of the performance, not only of the
headline workload): milliseconds, Mnodes/s and parked heavy bricks per launch."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import dgtest as T
    import discregrid_amd as dg
    dg.load_library()
    meshes = {"box (12)": T.box_mesh(), "torus (576)": T.torus(), "torus fine (57 600)": T.torus(240, 120),
              "icosphere nu=8 (1 280)": T.icosphere(8), "icosphere nu=71 (100 820)": T.icosphere(71),
              "bunny (69 630)": T.bunny_mesh()}
    s = torch.cuda.current_stream().cuda_stream
    for name, (V, F) in meshes.items():
        m = dg.Mesh(V, F)
        dom = dg.default_domain(V)
        sizes = [[64] * 3, [256] * 3, [512, 64, 32]]
        if len(F) >= 8192: # the meshes the kernel-choice rule (dg_capi.cpp) is about: more cell sizes around its threshold
            sizes += [[128] * 3, [384] * 3, [512] * 3]
        for res in sizes:
            g = dg.grid_desc(dom[:3], dom[3:], res)
            n = dg.n_nodes(g)
            out = torch.empty(n, dtype=torch.float64, device="cuda")
            ts = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                m.sample_nodes_device(g, 0, n, out.data_ptr(), stream=s)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            print("%-28s %-14s %9d nodes %9.3f ms %9.1f Mnodes/s  heavy %s" % (
                name, "x".join(map(str, res)), n, min(ts), n / min(ts) / 1e3, m.last_heavy_bricks()), flush=True)


if __name__ == "__main__":
    main()
