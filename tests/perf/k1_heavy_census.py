#!/usr/bin/env python3
"""How many bricks run out of their work budget (and how many of them the launch can park) on sphere-like and ordinary meshes:
dg_mesh_last_heavy_bricks after one K1 launch per (mesh, resolution), with the launch's time.  A launch that finds more heavy
bricks than it has slots lets the rest run on in their own wave -- the census shows how far the defaults (dg_kernels.h:
heavy_work_for, overflow_slots_for) are from that edge.   usage: python tests/perf/k1_heavy_census.py [res ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import dgtest as T
    import discregrid_amd as dg
    dg.load_library()
    dg.set_device(0)
    s = torch.cuda.current_stream().cuda_stream
    resolutions = [int(a) for a in sys.argv[1:]] or [128, 256]
    meshes = [("ico%d" % nu, (lambda nu=nu: T.icosphere(nu))) for nu in (25, 36, 50, 71, 100, 160)]
    meshes += [("torus", T.torus), ("bunny", T.bunny_mesh), ("dragon", T.dragon_mesh)]
    for name, make in meshes:
        V, F = make()
        mesh = dg.Mesh(V, F)
        dom = dg.default_domain(V)
        for r in resolutions:
            grid = dg.grid_desc(dom[:3], dom[3:], [r] * 3)
            n = dg.n_nodes(grid)
            buf = torch.empty(n, dtype=torch.float64, device="cuda")
            ts = []
            for _ in range(3):
                a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                mesh.sample_nodes_device(grid, 0, n, buf.data_ptr(), stream=s)
                z.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(z))
            heavy, split = mesh.last_heavy_bricks()
            print("%-8s %7d triangles  %4d^3: %8.3f ms  bricks over budget %6d, parked %6d%s" %
                  (name, len(F), r, min(ts), heavy, split, "   <-- more than the launch could park" if heavy > split else ""), flush=True)
            del buf


if __name__ == "__main__":
    main()
