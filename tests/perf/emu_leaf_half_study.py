"""Design study, round 6: would ONE bound per leaf half -- a rectangle / box bound for 4 triangles between the leaf's
own box and the per-pair rectangle bound of the float filter (FastWalk::leaf, dg_traverse.h) -- pay?  The CPU emulator runs the
product's traversal template and records, per visited leaf, which of its triangle pairs went past step 1 (the rectangle bound in the
triangle's own frame) for at least one lane.  A bound for a group of two pairs can skip the group's step-1 work only if NEITHER pair
reaches step 2 -- and no group bound is tighter than the pairs' own rectangles, so the count of such groups is an UPPER bound of what
any half-leaf bound could remove.  Kill criterion, written before the count: build only if >= 8 % fewer filter-pair steps on icosphere
AND bunny at 256^3.

    python tests/perf/emu_leaf_half_study.py [res] > profiles/r06_k1_leaf_half_study.txt
"""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes as C
import numpy as np
import dgtest as T
import emu

res = int(sys.argv[1]) if len(sys.argv) > 1 else 256
STEP1, STEP2, BOUND = 27, 44, 13          # VALU per pair at step 1 / step 2, per box bound (profiles/r05_k1_opcode_histogram.txt)
print("# leaf-half study on the emulator (product traversal template), %d^3, eight 4-plane slabs of the vertex class per mesh" % res)
print("# VALU model: step 1 %d per pair, step 2 %d per pair, one more bound %d; traversal total per brick 3 690 (icosphere, r05 histogram)" % (STEP1, STEP2, BOUND))
print("%-8s %9s %8s %8s %8s | %9s %9s | %9s %9s %9s | %s" % ("mesh", "leaves/br", "pairs1", "pairs2", "p1/leaf", "dead lvs", "their p1", "groups", "dead grp", "p1 saved", "net VALU per brick (upper bound of the gain)"))
for name, make in (("ico", lambda: T.icosphere(71)), ("bunny", T.bunny_mesh), ("dragon", T.dragon_mesh)):
    V, F = make()
    dom = T.oracle_default_domain(V)
    m = emu.EmuMesh(V, F)
    plane = (res + 1) ** 2
    runs = [(int(k) * 4 * plane, (int(k) * 4 + 4) * plane) for k in np.linspace(0, (res + 1) // 4 - 1, 8)]
    h = np.zeros(16, dtype=np.uint64)
    emu.lib().emu_leaf_half_stats(h.ctypes.data_as(C.c_void_p), 1)
    emu.set_fast(1)
    for b, e in runs:
        m.sample_range(dom, [res] * 3, b, e)
    B = emu.fast_stats()["bricks"]
    emu.lib().emu_leaf_half_stats(h.ctypes.data_as(C.c_void_p), 1)
    h = h.astype(np.float64)
    leaves, p1, p2, dead_l, dead_lp, groups, dead_g = h[:7]
    saved = 2.0 * dead_g                                   # pairs of step 1 a PERFECT group bound removes
    net = (saved * STEP1 - groups * BOUND) / B             # ... against one more bound per group visited
    print("%-8s %9.2f %8.2f %8.2f %8.2f | %9.2f %9.2f | %9.2f %9.2f %9.2f | %+.0f  (= %.1f %% of the step-1 pairs; sizes of the leaves in pairs 1..4: %s)"
          % (name, leaves / B, p1 / B, p2 / B, p1 / max(leaves, 1), dead_l / B, dead_lp / B, groups / B, dead_g / B, saved / B, net,
             100.0 * saved / max(p1, 1), " ".join("%.0f%%" % (100 * x / max(leaves, 1)) for x in h[8:12])))
