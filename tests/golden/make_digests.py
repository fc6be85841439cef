"""Pins 100 % of the lattice of the judged configurations, and BASELINE config 5 at its stated
size, to the UNMODIFIED reference (oracle/_ref/libdiscregrid_ref.so, `make -C oracle ref`; needs
/root/reference, i.e. runs in the build container only).

For every configuration the reference's own node loop (indexToNodePosition + signed_distance, what
CubicLagrangeDiscreteGrid::addFunction runs, cubic_lagrange_discrete_grid.cpp:806-831) is executed
over ALL nodes and the coefficient vector is reduced to one digest per block of 2^20 consecutive
nodes: the first 16 bytes of SHA-256 over the raw little-endian doubles.  The GPU tests
(tests/test_gpu_digests.py) reduce the device field the same way and compare every block.

  bunny128    BASELINE configs[1]: bunny (69 630 tris), 128^3, 14 926 977 nodes      (~1 min on 8 cores)
  ico71_256   BASELINE configs[2]: icosphere nu=71 (100 820 tris), 256^3, 118 425 857 nodes (~8 min)
  ico71_512   BASELINE configs[3]: same mesh, 512^3, 943 460 865 nodes               (~1 h)
  config5     BASELINE configs[4] on the ico71_256 field (taken from the reference run above):
              10 M points uniform in the domain (std::mt19937_64 seed 1234) and 10 M points of the
              SPH-like shell |phi| < 2h, h = 0.1 (the first 10 M of a 26 M-point uniform stream,
              seed 4321, whose reference value passes the test); CubicLagrangeDiscreteGrid::
              interpolate value-only and value+gradient (:977-1063); digests per 2^20 queries.

Output: tests/golden/lattice_digests.npz (a few tens of KB).

Run:  python tests/golden/make_digests.py [bunny128] [ico71_256] [ico71_512]     (default: all)
"""
import os
import sys
import time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import dgtest as T  # noqa: E402

OUT = os.path.join(HERE, "lattice_digests.npz")
BLOCK = 1 << 20
CHUNK = 1 << 22           # nodes per call into the reference (whole blocks)
N_QUERIES = 10_000_000
SHELL_STREAM = 26_000_000
SHELL_H = 0.1
DBL_MAX = np.finfo(np.float64).max


def meshes():
    return {
        "bunny128": (T.bunny_mesh, [128, 128, 128]),
        "ico71_256": (lambda: T.icosphere(71), [256, 256, 256]),
        "ico71_512": (lambda: T.icosphere(71), [512, 512, 512]),
    }


def lattice(name, keep_field):
    make, res = meshes()[name]
    V, F = make()
    dom = T.ref_default_domain(V)
    g = T.RefGrid(V, F, dom, res)
    n = T.n_nodes(res)
    dig = np.empty(((n + BLOCK - 1) // BLOCK, 16), dtype=np.uint8)
    field = np.empty(n) if keep_field else None
    t0 = time.time()
    for b in range(0, n, CHUNK):
        e = min(n, b + CHUNK)
        vals = g.sample_nodes(b, e)
        dig[b // BLOCK:(e + BLOCK - 1) // BLOCK] = T.block_digests(vals, BLOCK)
        if keep_field:
            field[b:e] = vals
        if (b // CHUNK) % 8 == 0:
            print("%s: %d / %d nodes, %.0f s" % (name, e, n, time.time() - t0), flush=True)
    out = {name + "_domain": dom, name + "_res": np.array(res, dtype=np.uint32), name + "_nodes": np.uint64(n),
           name + "_digest": dig, name + "_seconds": np.float64(time.time() - t0)}
    return out, (V, F, dom, res, field)


def config5(dom, res, field):
    g = T.RefGrid(None, None, dom, res)
    g.add_coeffs(field)
    out = {}
    P = T.uniform_points(1234, N_QUERIES, dom[:3], dom[3:])
    phi = g.interpolate(P)
    phi_g, grad = g.interpolate(P, grad=True)
    grad[phi_g == DBL_MAX] = 0.0            # the reference leaves it uninitialised there
    out["c5_uniform_phi"] = T.block_digests(phi, BLOCK)
    out["c5_uniform_phi_g"] = T.block_digests(phi_g, BLOCK)
    out["c5_uniform_grad"] = T.block_digests(grad, BLOCK)
    out["c5_uniform_phi_head"] = phi[:64].copy()
    C = T.uniform_points(4321, SHELL_STREAM, dom[:3], dom[3:])
    phic = g.interpolate(C)
    keep = np.flatnonzero((phic != DBL_MAX) & (np.abs(phic) < 2 * SHELL_H))
    assert len(keep) >= N_QUERIES, "shell stream too short: %d" % len(keep)
    S = C[keep[:N_QUERIES]]
    phi = g.interpolate(S)
    phi_g, grad = g.interpolate(S, grad=True)
    grad[phi_g == DBL_MAX] = 0.0
    out["c5_shell_points"] = T.block_digests(S, BLOCK)
    out["c5_shell_phi"] = T.block_digests(phi, BLOCK)
    out["c5_shell_phi_g"] = T.block_digests(phi_g, BLOCK)
    out["c5_shell_grad"] = T.block_digests(grad, BLOCK)
    out["c5_shell_last_candidate"] = np.uint64(keep[N_QUERIES - 1])
    out["c5_n_queries"] = np.uint64(N_QUERIES)
    out["c5_shell_stream"] = np.uint64(SHELL_STREAM)
    return out


def main():
    assert T.ref_available(), "build oracle/_ref first: make -C oracle ref"
    want = sys.argv[1:] or list(meshes())
    res = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    res["block"] = np.uint64(BLOCK)
    for name in want:
        out, (V, F, dom, r, field) = lattice(name, keep_field=(name == "ico71_256"))
        res.update(out)
        np.savez_compressed(OUT, **res)
        if name == "ico71_256":
            res.update(config5(dom, r, field))
            np.savez_compressed(OUT, **res)
            del field
        print("%s done" % name, flush=True)


if __name__ == "__main__":
    main()
