"""Recomputes the ALGORITHMIC bytes per node of SURVEY.md section 8(d) for the headline
workload over the FULL 256^3 lattice (118 425 857 nodes) with the oracle's instrumented
reference traversal, and freezes it in profiles/balg_icosphere71_256.json:

    B_alg = Vbar * 72 + Lbar * 84 + 8   bytes/node
    Vbar / Lbar = mean inner-node / leaf visits of the reference's _query
                  (TriangleMeshDistance.h:514-562); 72 = sizeof(Node), 84 = 12-byte index
                  triple + 3 x 24-byte vertices, 8 = the coefficient store.

Takes ~8 minutes on 8 cores.  Run: python tests/count_reference_visits.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dgtest as T  # noqa: E402


def main():
    V, F = T.icosphere(71)
    om = T.OracleMesh(V, F)
    dom = T.oracle_default_domain(V)
    res = [256] * 3
    n = T.n_nodes(res)
    tot = np.zeros(2, dtype=np.uint64)
    secs = 0.0
    t0 = time.time()
    for b in range(0, n, 4_000_000):
        e = min(n, b + 4_000_000)
        _, vis = om.sample_nodes(dom, res, b, e, visits=True)
        tot += vis
        secs += om.last_seconds
        print(b, e, vis.tolist(), round(om.last_seconds, 2), flush=True)
    vbar, lbar = float(tot[0]) / n, float(tot[1]) / n
    out = {
        "workload": "icosphere nu=71 (100820 tris), 256^3 grid, default domain, all %d nodes" % n,
        "vbar_inner_visits_per_node": vbar,
        "lbar_leaf_visits_per_node": lbar,
        "bytes_per_node": vbar * 72 + lbar * 84 + 8,
        "oracle_cpu_seconds": secs, "oracle_threads": os.cpu_count(),
        "oracle_mnodes_per_s": n / secs / 1e6,
    }
    p = os.path.join(T.ROOT, "profiles", "balg_icosphere71_256.json")
    old = {}
    if os.path.exists(p):
        old = json.load(open(p))
    for k in ("measured_hbm_bytes_per_launch", "measured_hbm_note"):
        if k in old:
            out[k] = old[k]
    json.dump(out, open(p, "w"), indent=2)
    print(json.dumps(out), "wall %.0fs" % (time.time() - t0))


if __name__ == "__main__":
    main()
