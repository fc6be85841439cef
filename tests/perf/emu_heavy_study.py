"""Design study (CPU emulator, the product's own traversal template): what do the jobs of k_heavy_subtrees cost as they run today
(the EXACT walk of one top-level subtree per wave: the double test on every triangle some lane's box bound reaches), and what would
they cost with the FILTERED walk seeded by the parked upper bounds (float filter, per-lane candidate lists, the double test on the
candidates only; a job in which some lane's list fills up falls back to the exact walk)?
Slabs of the vertex class through the centre of the mesh, where the heavy bricks are.
usage: python tests/perf/emu_heavy_study.py [ico|bunny|dragon] [res]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import dgtest as T
import emu

mesh = sys.argv[1] if len(sys.argv) > 1 else "ico"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 256
V, F = {"ico": lambda: T.icosphere(71), "bunny": T.bunny_mesh, "dragon": T.dragon_mesh}[mesh]()
dom = T.oracle_default_domain(V)
R = [res] * 3
m = emu.EmuMesh(V, F)
plane = (res + 1) ** 2
mid = (res + 1) // 8 * 4
L = emu.lib()
L.emu_set_heavy_study(1)
emu.set_fast(1)
heavy = 0
for k in range(mid - 16, mid + 16, 4):       # eight 4-plane slabs around the middle
    m.sample_range(dom, R, k * plane, (k + 4) * plane)
    heavy += int(m.stats["heavy_bricks"])
h = (C.c_uint64 * 16)()
L.emu_heavy_study(h)
h = [int(x) for x in h]
hh = (C.c_uint64 * 8)()
L.emu_heavy_study_hist(hh)
print("  jobs by double tests (0, 1-9, 10-49, 50-99, 100-199, 200-299, 300-379, 380+):", [int(x) for x in hh])
L.emu_set_heavy_study(0)
jobs = max(h[0], 1)
# vector instructions per event (gfx950 ISA of the two walks: profiles/r05_k1_opcode_histogram.txt; the exact walk's pair step carries
# the per-slab error term, its leaf tests a bound pair per two triangles and the double test per triangle some lane needs)
EX = {"pair": 38, "group": 36, "test": 125}
FA = {"pair": 34, "leaf": 14, "s1": 27, "s2": 53, "round": 115, "job": 120}
exact = h[1] * EX["pair"] + h[2] * EX["group"] + h[3] * EX["test"]
fast = (h[4] * FA["pair"] + h[5] * FA["leaf"] + h[6] * FA["s1"] + h[7] * FA["s2"] + h[9] * FA["round"] + (h[0] - 0) * FA["job"]
        + h[11] * EX["pair"] + h[12] * EX["group"] + h[13] * EX["test"])
print("%s %d^3, vertex planes %d..%d: %d heavy bricks, %d jobs (%d pruned at their root pair)" % (mesh, res, mid - 16, mid + 16, heavy, h[0], h[14]))
print("  exact walk   per job: pair steps %.2f  bound pairs %.2f  double tests %.2f   -> %.0f vector instructions" %
      (h[1] / jobs, h[2] / jobs, h[3] / jobs, exact / jobs))
print("  filtered walk per job: pair steps %.2f  leaf visits %.2f  filter pairs step 1 %.2f  step 2 %.2f  candidates %.2f  pooled rounds %.2f" %
      (h[4] / jobs, h[5] / jobs, h[6] / jobs, h[7] / jobs, h[8] / jobs, h[9] / jobs))
print("  jobs that fall back to the exact walk (a list filled up): %d = %.1f %%, carrying %.1f %% of the exact walk's double tests" %
      (h[10], 100.0 * h[10] / jobs, 100.0 * h[13] / max(h[3], 1)))
print("  modelled vector instructions per job: exact %.0f, filtered + fallbacks %.0f  (%.2f x)" % (exact / jobs, fast / jobs, fast / max(exact, 1)))
