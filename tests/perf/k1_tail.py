#!/usr/bin/env python3
"""How long does the slowest brick of the headline workload run?  Times K1 on thin slices of
the vertex-class lattice (planes around a given k), each small enough to fit the GPU in one
wave of blocks, so the launch duration ~ the longest brick in the slice."""
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
    V, F = T.icosphere(71)
    dom = dg.default_domain(V)
    grid = dg.grid_desc(dom[:3], dom[3:], [256] * 3)
    mesh = dg.Mesh(V, F)
    n = dg.n_nodes(grid)
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    plane = 257 * 257

    def run(b, e, reps=3):
        ts = []
        for _ in range(reps):
            a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            mesh.sample_nodes_device(grid, b, e, out.data_ptr(), stream=s)
            z.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(z))
        return min(ts)

    print("full launch: %.2f ms" % run(0, n))
    for k in (0, 32, 64, 96, 112, 120, 124, 128, 132, 160, 200, 252):
        print("V planes %3d..%3d (%d bricks): %.3f ms" % (k, k + 4, 65 * 65, run(k * plane, (k + 4) * plane)))
    # a single brick row through the centre
    for j in (0, 64, 120, 128):
        b = 128 * plane + j * 257
        print("V plane 128, rows %3d..%3d: %.3f ms" % (j, j + 4, run(b, b + 4 * 257)))


if __name__ == "__main__":
    main()
