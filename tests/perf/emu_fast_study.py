"""Design study for the filtered K1 traversal (CPU emulator): work counters, candidate-list statistics and
bit parity against the exact traversal.  usage: python tests/perf/emu_fast_study.py [nu] [res] [cap] [leaf]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import dgtest as T
import emu

nu = int(sys.argv[1]) if len(sys.argv) > 1 else 71
res = int(sys.argv[2]) if len(sys.argv) > 2 else 256
cap = 8  # kFastListCap (compile-time)
leaf = int(sys.argv[4]) if len(sys.argv) > 4 else 6
mesh = sys.argv[5] if len(sys.argv) > 5 else "ico"
if mesh == "ico":
    V, F = T.icosphere(nu)
elif mesh == "bunny":
    V, F = T.bunny_mesh()
elif mesh == "torus":
    V, F = T.torus(nu, nu // 2)
dom = T.oracle_default_domain(V)
R = [res] * 3
N = T.n_nodes(R)
m = emu.EmuMesh(V, F, max_leaf=leaf)
# V-class nodes first in the order: take evenly spread runs of whole 4-plane slabs of the vertex class
nv = (res + 1) ** 3
plane = (res + 1) ** 2
runs = [(int(k) * 4 * plane, (int(k) * 4 + 4) * plane) for k in np.linspace(0, (res + 1) // 4 - 1, 8)]
tot = dict()
bad = 0
t0 = time.time()
for (b, e) in runs:
    emu.set_fast(0)
    ref = m.sample_range(dom, R, b, e)
    st0 = dict(m.stats)
    emu.set_fast(1)
    got = m.sample_range(dom, R, b, e)
    fs = emu.fast_stats()
    emu.set_fast(0)
    bad += int((ref.view(np.uint64) != got.view(np.uint64)).sum())
    for k, v in fs.items():
        if k == "hist":
            tot[k] = np.array(v) + tot.get(k, 0)
        else:
            tot[k] = tot.get(k, 0) + v
    for k, v in st0.items():
        tot["x_" + k] = tot.get("x_" + k, 0) + v
B = tot["bricks"]
print("mesh %s nu=%d tris=%d res=%d cap=%d leaf=%d  bricks=%d  bit mismatches=%d  (%.0f s)" % (mesh, nu, len(F), res, cap, leaf, B, bad, time.time() - t0))
print("exact : pair_steps/brick %.1f  leaf_visits %.1f  tri-pair bounds %.1f  exact tests %.1f" % (
    tot["x_node_visits"] / 2 / tot["x_bricks"], tot["x_leaf_visits"] / tot["x_bricks"], tot["x_leaf_groups"] / tot["x_bricks"], tot["x_tri_tests"] / tot["x_bricks"]))
print("fast  : pair_steps/brick %.1f  leaf_visits %.1f  tri pairs %.1f  appends/lane %.2f resets/lane %.2f" % (
    tot["pair_steps"] / B, tot["leaf_visits"] / B, tot["tri_pairs"] / B, tot["appends"] / max(1, tot["lanes"]), tot["resets"] / max(1, tot["lanes"])))
print("        overflow bricks %d (%.3f%%)  mean list %.2f  mean max-list/brick %.2f" % (
    tot["redo_bricks"], 100.0 * tot["redo_bricks"] / B, tot["sum_list"] / max(1, tot["lanes"]), tot["sum_max_list"] / max(1, B - tot["redo_bricks"])))
print("        list length histogram:", (tot["hist"] / max(1, tot["hist"].sum())).round(4).tolist())
cur = 40 * tot["x_node_visits"] / 2 / tot["x_bricks"] + 34 * tot["x_leaf_groups"] / tot["x_bricks"] + 71 * tot["x_tri_tests"] / tot["x_bricks"]
new = 30 * tot["pair_steps"] / B + 64 * tot["tri_pairs"] / B + 90 * tot["sum_max_list"] / max(1, B) + 15 * tot["leaf_visits"] / B
print("VALU model per brick: exact %.0f  fast %.0f  (ratio %.2f)" % (cur, new, new / cur))
try:
    o = np.zeros(2, dtype=np.uint64)
    emu.lib().emu_need(o.ctypes.data_as(T.c_u64p))
    print("needed (d2 <= U_final): mean/lane %.2f  max/brick %.2f" % (o[0] / max(1, tot["lanes"]), o[1] / max(1, B)))
except AttributeError:
    pass
