"""Golden density maps on a deeper grid than the other density fixtures (which are 6 cells deep):
torus SDF 12 x 11 x 9 from the UNMODIFIED reference (oracle/_ref), density map for a small and a
large support radius (h = 0.45 is two cells: the quadrature box spans four cells per axis).
Writes tests/golden/density_box.npz.

Run:  python tests/golden/make_golden_density_box.py
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import dgtest as T  # noqa: E402


def main():
    assert T.ref_available(), "build oracle/_ref first: make -C oracle ref"
    V, F = T.torus()
    dom = T.ref_default_domain(V)
    res = [12, 11, 9]
    out = {"domain": dom, "res": np.array(res, dtype=np.uint32)}
    g = T.RefGrid(V, F, dom, res)
    g.add_sdf()
    out["sdf"] = g.nodes(0)
    for tag, h in (("h012", 0.12), ("h045", 0.45)):
        g2 = T.RefGrid(V, F, dom, res)
        g2.add_sdf()
        secs = g2.add_density_map(h, 1000.0)
        print("reference density map h=%g: %.1f s" % (h, secs))
        out["density_" + tag] = g2.nodes(1)
    np.savez_compressed(os.path.join(HERE, "density_box.npz"), **out)
    print("wrote density_box.npz")


if __name__ == "__main__":
    main()
