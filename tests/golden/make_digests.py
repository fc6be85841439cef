"""Pins 100 % of the lattice of the judged configurations, and BASELINE config 5 at its stated
size, to the UNMODIFIED reference (oracle/_ref/libdiscregrid_ref.so, `make -C oracle ref`; needs
/root/reference, i.e. runs in the build container only).

For every configuration the reference's own node loop (indexToNodePosition + signed_distance, what
CubicLagrangeDiscreteGrid::addFunction runs, cubic_lagrange_discrete_grid.cpp:806-831) is executed
over ALL nodes and the coefficient vector is reduced to one digest per block of 2^20 consecutive
nodes: the first 16 bytes of SHA-256 over the raw little-endian doubles.  The GPU tests
(tests/test_gpu_digests.py) reduce the device field the same way and compare every block.

  bunny128    BASELINE configs[1]: bunny (69 630 tris), 128^3, 14 926 977 nodes      (~1 min on 8 cores)
  dragon128   the reference's third sample mesh, cmd/generate_sdf/resources/dragon.obj (79 988 tris), 128^3 (~1 min; also
              stages the mesh as tests/golden/dragon.npz -- the OBJ only exists under /root/reference)
  ico224_sample  a ONE-MILLION-triangle icosphere (nu = 224: 1 003 520 tris), 128^3: 20 011 lattice nodes, every 746-th, through
              the reference's node loop (values, not digests: tests/golden/big_mesh_sample.npz)
  ico71_256   BASELINE configs[2]: icosphere nu=71 (100 820 tris), 256^3, 118 425 857 nodes (~8 min)
  ico71_512   BASELINE configs[3]: same mesh, 512^3, 943 460 865 nodes               (~1 h)
  config5     BASELINE configs[4] on the ico71_256 field (taken from the reference run above):
              10 M points uniform in the domain (std::mt19937_64 seed 1234) and 10 M points of the
              SPH-like shell |phi| < 2h, h = 0.1 (the first 10 M of a 26 M-point uniform stream,
              seed 4321, whose reference value passes the test); CubicLagrangeDiscreteGrid::
              interpolate value-only and value+gradient (:977-1063); digests per 2^20 queries.

  density128  K3 at full lattice: the icosphere SDF at 128^3 (14 926 977 nodes, sampled by the reference; its
              digests are stored as ico71_128_*) goes through the UNMODIFIED reference's GenerateDensityMap
              lambdas -- node predicate + density_func = 1 + 4096 interpolate calls per integrated node,
              cmd/generate_density_map/main.cpp:86-133, h = 0.1, rho0 = 1000, band predicate on -- over ALL
              nodes (46 G interpolations, ~1 h on 8 cores; resumable: partial results are kept in
              /tmp/dg_density128_partial.npz).  Digests per 2^20 nodes + a strided sample of the values.
  density256  the same on the 256^3 field of ico71_256 (BASELINE configs[4]'s SDF): all 118 425 857 nodes, 89.7 M
              integrated = 367 G interpolate calls of the UNMODIFIED reference (~2 h on 8 cores; resumable).

Output: tests/golden/lattice_digests.npz (a few tens of KB).

Run:  python tests/golden/make_digests.py [bunny128] [ico71_256] [ico71_512] [density128] [density256]
      (default: the three lattices)
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
        "dragon128": (stage_dragon, [128, 128, 128]),
        "ico71_256": (lambda: T.icosphere(71), [256, 256, 256]),
        "ico71_512": (lambda: T.icosphere(71), [512, 512, 512]),
    }


def stage_dragon():
    path = os.path.join(HERE, "dragon.npz")
    if not os.path.exists(path):
        V, F = T.load_obj("/root/reference/cmd/generate_sdf/resources/dragon.obj")
        np.savez_compressed(path, V=V, F=F)
    return T.dragon_mesh()


def ico224_sample():
    """a million triangles: strided lattice nodes through the unmodified reference, values kept"""
    V, F = T.icosphere(224)
    res = [128, 128, 128]
    dom = T.ref_default_domain(V)
    g = T.RefGrid(V, F, dom, res)
    n = T.n_nodes(res)
    idx = np.arange(0, n, 746, dtype=np.uint64)
    t0 = time.time()
    sd = np.array([g.sample_nodes(int(l), int(l) + 1)[0] for l in idx])
    np.savez_compressed(os.path.join(HERE, "big_mesh_sample.npz"), nu=np.uint32(224), triangles=np.uint64(len(F)), res=np.array(res, dtype=np.uint32),
                        domain=dom, idx=idx, sd=sd, seconds=np.float64(time.time() - t0))
    print("ico224_sample: %d nodes of %d, %d triangles, %.0f s" % (len(idx), n, len(F), time.time() - t0), flush=True)


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


DENSITY_H = 0.1
DENSITY_RHO0 = 1000.0
DENSITY_SAMPLE_STRIDE = 4099


def density_full(tag, n_res):
    """K3's reference at full lattice, resumable: the icosphere SDF at n_res^3 sampled by the reference, then the
    reference's GenerateDensityMap lambdas over every node."""
    V, F = T.icosphere(71)
    res = [n_res] * 3
    dom = T.ref_default_domain(V)
    n = T.n_nodes(res)
    part_path = "/tmp/dg_%s_partial.npz" % tag
    part = dict(np.load(part_path)) if os.path.exists(part_path) else {}
    if "sdf" in part:
        sdf = part["sdf"]
    else:
        g = T.RefGrid(V, F, dom, res)
        sdf = np.empty(n)
        for b in range(0, n, CHUNK):
            e = min(n, b + CHUNK)
            sdf[b:e] = g.sample_nodes(b, e)
        del g
        part = {"sdf": sdf, "done": np.uint64(0), "dens": np.empty(n)}
        np.savez(part_path, **part)
    g = T.RefGrid(None, None, dom, res)
    assert g.add_coeffs(sdf) == 0
    dens = part["dens"]
    done = int(part["done"])
    step = 1 << 18
    t0 = time.time()
    secs = float(part.get("seconds", 0.0))
    for b in range(done, n, step):
        e = min(n, b + step)
        dens[b:e] = g.density_nodes(DENSITY_H, DENSITY_RHO0, b, e)
        secs += g.last_seconds
        part.update(done=np.uint64(e), dens=dens, seconds=np.float64(secs))
        if (b // step) % 16 == 15 or e == n:
            np.savez(part_path, **part)
        print("%s: %d / %d nodes, %.0f s (this run), %.0f s total" % (tag, e, n, time.time() - t0, secs), flush=True)
    integrated = int(np.count_nonzero((dens != DBL_MAX) & (sdf <= 2 * DENSITY_H)))
    sdf_tag = "ico71_%d" % n_res
    out = {tag + "_h": np.float64(DENSITY_H), tag + "_rho0": np.float64(DENSITY_RHO0),
           tag + "_digest": T.block_digests(dens, BLOCK),
           tag + "_sample_stride": np.uint64(DENSITY_SAMPLE_STRIDE),
           tag + "_sample": dens[::DENSITY_SAMPLE_STRIDE].copy(),
           tag + "_integrated_nodes": np.uint64(integrated),
           tag + "_seconds": np.float64(secs)}
    if n_res != 256:       # (ico71_256_* is written by the lattice run)
        out.update({sdf_tag + "_domain": dom, sdf_tag + "_res": np.array(res, dtype=np.uint32), sdf_tag + "_nodes": np.uint64(n),
                    sdf_tag + "_digest": T.block_digests(sdf, BLOCK)})
    return out


def density128():
    return density_full("density128", 128)


def density256():
    return density_full("density256", 256)


def main():
    assert T.ref_available(), "build oracle/_ref first: make -C oracle ref"
    want = sys.argv[1:] or ["bunny128", "ico71_256", "ico71_512"]
    if "ico224_sample" in want:
        want.remove("ico224_sample")
        ico224_sample()
    res = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    res["block"] = np.uint64(BLOCK)
    for name, fn in (("density128", density128), ("density256", density256)):
        if name in want:
            want.remove(name)
            res.update(fn())
            np.savez_compressed(OUT, **res)
            print("%s done" % name, flush=True)
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
