#!/usr/bin/env python3
"""End-to-end time of the HOST-pointer entry point dg_sdf_sample_nodes (what the C++ addFunction
calls): device allocation + K1 + D2H into the caller's pageable memory."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import dgtest as T
    import discregrid_amd as dg
    dg.load_library()
    V, F = T.icosphere(71)
    dom = dg.default_domain(V)
    for res in ([128] * 3, [256] * 3):
        grid = dg.grid_desc(dom[:3], dom[3:], res)
        mesh = dg.Mesh(V, F)
        n = dg.n_nodes(grid)
        out = None
        for rep in range(4):
            del out              # (freeing 1 GB takes tens of ms: keep it out of the timed region)
            t0 = time.perf_counter()
            out = mesh.sample_nodes(grid)
            dt = time.perf_counter() - t0
            print("%s rep %d: host call %.1f ms (kernel %.1f ms) = %.0f Mnodes/s end to end, checksum %.6f" % (
                "x".join(map(str, res)), rep, dt * 1e3, dg.last_kernel_ms(), n / dt / 1e6, float(out[::9973].sum())), flush=True)


def others():
    """Host-pointer K2 (10 M queries, value + gradient) and K1p (2 M points) end to end."""
    import dgtest as T
    import discregrid_amd as dg
    V, F = T.icosphere(71)
    dom = dg.default_domain(V)
    grid = dg.grid_desc(dom[:3], dom[3:], [128] * 3)
    mesh = dg.Mesh(V, F)
    coeffs = mesh.sample_nodes(grid)
    field = dg.Field(grid, coeffs)
    rng = np.random.default_rng(2)
    Q = rng.uniform(dom[:3], dom[3:], size=(10_000_000, 3))
    for grad in (False, True):
        for rep in range(3):
            t0 = time.perf_counter()
            out = field.interpolate(Q, grad=grad)
            dt = time.perf_counter() - t0
        print("interpolate host, 10 M queries, grad=%s: %.1f ms (device part %.1f ms) = %.0f Mq/s" % (
            grad, dt * 1e3, dg.last_kernel_ms(), len(Q) / dt / 1e6), flush=True)
        del out
    P = Q[:2_000_000]
    for rep in range(3):
        t0 = time.perf_counter()
        d = mesh.signed_distance(P)
        dt = time.perf_counter() - t0
    print("signed_distance host, 2 M points: %.1f ms (device part %.1f ms) = %.0f Mpoints/s" % (
        dt * 1e3, dg.last_kernel_ms(), len(P) / dt / 1e6), flush=True)


if __name__ == "__main__":
    main()
    others()
