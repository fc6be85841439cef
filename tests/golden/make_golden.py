"""Generates the committed golden fixtures under tests/golden/ from the UNMODIFIED
reference (oracle/_ref/libdiscregrid_ref.so, built by `make -C oracle ref`; needs
/root/reference, i.e. runs in the build container only):

  box.cdf          byte copy of the reference's only golden file
                   (cmd/generate_sdf/resources/box.cdf = GenerateSDF -r "5 5 5" box.obj)
  bunny.npz        vertices/faces of cmd/generate_sdf/resources/bunny.obj as parsed by the
                   reference's OBJ subset (the OBJ does not exist on the GPU box)
  ref_vectors.npz  outputs of the reference on seeded inputs: SDF coefficients
                   (addFunction), signed_distance results, interpolate values+gradients,
                   node positions, for box / icosphere / torus / bunny.

Run:  python tests/golden/make_golden.py
"""
import os
import shutil
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import dgtest as T  # noqa: E402

RES_DIR = os.path.join(T.REF_ROOT, "cmd", "generate_sdf", "resources")


def main():
    assert T.ref_available(), "build oracle/_ref first: make -C oracle ref"
    shutil.copyfile(os.path.join(RES_DIR, "box.cdf"), os.path.join(HERE, "box.cdf"))
    Vb, Fb = T.load_obj(os.path.join(RES_DIR, "bunny.obj"))
    np.savez_compressed(os.path.join(HERE, "bunny.npz"), V=Vb, F=Fb)

    out = {}
    cases = {
        "box": (T.box_mesh(), [32, 32, 32]),        # BASELINE config 1 (245 025 nodes)
        "ico8": (T.icosphere(8), [12, 10, 11]),
        "torus": (T.torus(), [9, 14, 6]),
        "bunny": ((Vb, Fb), [16, 16, 16]),
    }
    rng = np.random.default_rng(20260925)
    for name, ((V, F), res) in cases.items():
        dom = T.ref_default_domain(V)
        g = T.RefGrid(V, F, dom, res)
        g.add_sdf()
        coeffs = g.nodes()
        ext = dom[3:] - dom[:3]
        P = rng.uniform(dom[:3] - 0.1 * ext, dom[3:] + 0.1 * ext, size=(4096, 3))
        P[:8] = dom[3:]                       # domain.max corner (inclusive contains)
        P[8:16] = dom[:3]
        # points on cell faces
        hdr = g.header()
        ijk = rng.integers(0, np.array(res) + 1, size=(64, 3))
        P[16:80] = dom[:3] + ijk * hdr["cell"]
        d, tri, ent, near = g.signed_distance(P, full=True)
        phi, grad = g.interpolate(P, grad=True)
        inside = phi != np.finfo(np.float64).max
        grad[~inside] = 0.0                   # reference leaves it uninitialised there
        out[name + "_domain"] = dom
        out[name + "_res"] = np.array(res, dtype=np.uint32)
        out[name + "_coeffs"] = coeffs
        out[name + "_P"] = P
        out[name + "_sd"] = d
        out[name + "_tri"] = tri
        out[name + "_ent"] = ent
        out[name + "_near"] = near
        out[name + "_phi"] = phi
        out[name + "_grad"] = grad
        sel = rng.integers(0, len(coeffs), size=512).astype(np.uint32)
        out[name + "_pos_idx"] = sel
        out[name + "_pos"] = np.stack([g.node_positions(int(l), int(l) + 1)[0] for l in sel])
        print(name, len(F), "tris", len(coeffs), "nodes")
    # inverted SDF (cmd/generate_sdf/main.cpp:95-98)
    V, F = T.torus()
    dom = T.ref_default_domain(V)
    g = T.RefGrid(V, F, dom, [7, 7, 7])
    g.add_sdf(invert=True)
    out["torus_inv_coeffs"] = g.nodes()
    # config-3 mesh (icosphere nu=71, 100 820 tris): signed distance at seeded points and a
    # strided sample of the 256^3 lattice (every 99 991-th node)
    V, F = T.icosphere(71)
    dom = T.ref_default_domain(V)
    res = [256, 256, 256]
    g = T.RefGrid(V, F, dom, res)
    P = rng.uniform(dom[:3], dom[3:], size=(2048, 3))
    P[0] = 0.0
    out["ico71_domain"] = dom
    out["ico71_P"] = P
    out["ico71_sd"] = g.signed_distance(P)
    idx = np.arange(0, T.n_nodes(res), 99991, dtype=np.uint32)
    out["ico71_lattice_idx"] = idx
    out["ico71_lattice_sd"] = np.array([g.sample_nodes(int(l), int(l) + 1)[0] for l in idx])
    # bunny 128^3 (config 2) strided lattice sample
    dom = T.ref_default_domain(Vb)
    res = [128, 128, 128]
    g = T.RefGrid(Vb, Fb, dom, res)
    idx = np.arange(0, T.n_nodes(res), 9973, dtype=np.uint32)
    out["bunny128_domain"] = dom
    out["bunny128_lattice_idx"] = idx
    out["bunny128_lattice_sd"] = np.array([g.sample_nodes(int(l), int(l) + 1)[0] for l in idx])
    # reduceField golden (cubic_lagrange_discrete_grid.cpp:1065-1174): box.cdf reduced with |v| < 0.25
    g = T.RefGrid(path=os.path.join(HERE, "box.cdf"))
    g.reduce_abs_lt(0, 0.25)
    g.save(os.path.join(HERE, "box_reduced_0p25.cdf"))
    # and a bigger, Morton-sorted one: torus 9x14x6, |v| < 0.08
    V, F = T.torus()
    dom = T.ref_default_domain(V)
    g = T.RefGrid(V, F, dom, [9, 14, 6])
    g.add_sdf()
    g.save(os.path.join(HERE, "torus_9_14_6.cdf"))
    g.reduce_abs_lt(0, 0.08)
    g.save(os.path.join(HERE, "torus_9_14_6_reduced_0p08.cdf"))
    # density map (cmd/generate_density_map): torus SDF 9x14x6, h = 0.15, rho0 = 1000
    gx, gw = T.parse_reference_gauss_rule(30)
    assert len(gx) == 16 and len(gw) == 16
    out["gauss16_x"], out["gauss16_w"] = gx, gw
    g = T.RefGrid(path=os.path.join(HERE, "torus_9_14_6.cdf"))
    secs = g.add_density_map(0.15, 1000.0)
    print("density map (reference): %.1f s" % secs)
    out["torus_density_h015"] = g.nodes(1)
    g.save(os.path.join(HERE, "torus_9_14_6_density.cdm"))         # = GenerateDensityMap --no-reduction
    g.reduce_density(0.15, 1000.0)
    g.save(os.path.join(HERE, "torus_9_14_6_density_reduced.cdm")) # = GenerateDensityMap
    # finer lattice where the node predicate and both reductions actually remove something
    V, F = T.torus()
    dom = T.ref_default_domain(V)
    g = T.RefGrid(V, F, dom, [16, 16, 6])
    g.add_sdf()
    g.save(os.path.join(HERE, "torus_16_16_6.cdf"))
    g.add_density_map(0.1, 1000.0)
    out["torus16_density_h01"] = g.nodes(1)
    g.reduce_density(0.1, 1000.0)
    g.save(os.path.join(HERE, "torus_16_16_6_density_reduced.cdm"))
    g2 = T.RefGrid(path=os.path.join(HERE, "torus_9_14_6.cdf"))
    g2.add_density_map(0.15, 1000.0, no_reduction=True)
    out["torus_density_h015_nopred"] = g2.nodes(1)
    np.savez_compressed(os.path.join(HERE, "ref_vectors.npz"), **out)
    print("wrote", os.path.join(HERE, "ref_vectors.npz"))


if __name__ == "__main__":
    main()
