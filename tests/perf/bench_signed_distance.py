#!/usr/bin/env python3
"""K1p: batched TriangleMeshDistance::signed_distance at caller-supplied points (icosphere nu=71).
Lattice-ordered points (what K1 sees), the same points shuffled, and uniform random points."""
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
    mesh = dg.Mesh(V, F)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    g = torch.Generator(device="cuda")
    g.manual_seed(7)
    k = int(round(n ** (1 / 3)))
    ax = torch.linspace(-1.2, 1.2, k, dtype=torch.float64, device="cuda")
    lattice = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3).contiguous()
    shuffled = lattice[torch.randperm(len(lattice), device="cuda", generator=g)].contiguous()
    uniform = (torch.rand((len(lattice), 3), dtype=torch.float64, device="cuda", generator=g) * 2.4 - 1.2).contiguous()
    out = torch.empty(len(lattice), dtype=torch.float64, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for name, pts in (("lattice order", lattice), ("lattice shuffled", shuffled), ("uniform random", uniform)):
        m = len(pts)
        for _ in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            mesh.signed_distance_device(pts.data_ptr(), m, out.data_ptr(), stream=s)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print("%-18s %9d points: %9.2f ms  %8.1f Mpoints/s  checksum %.6f" % (name, m, ms, m / ms / 1e3, float(out[::997].sum())), flush=True)


if __name__ == "__main__":
    main()
