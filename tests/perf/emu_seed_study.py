"""Design study (CPU emulator, the product's own traversal template): what would a SEEDED upper bound save the filtered K1
traversal?  Every brick is walked twice: as the kernel does (U = +inf at the root), and starting from an ORACLE-TIGHT upper bound
per lane (the lane's final U of a first, uncounted traversal) -- the best any coarse pre-pass over super-bricks could hand down --
and from what a pre-pass that samples the brick CENTRES could really give: (own distance + two brick radii)^2.
Counted: node-pair steps, leaf visits, filter pairs (step 1 / step 2), list appends per brick.
usage: python tests/perf/emu_seed_study.py [ico|bunny|dragon] [res]"""
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
runs = [(int(k) * 4 * plane, (int(k) * 4 + 4) * plane) for k in np.linspace(0, (res + 1) // 4 - 1, 8)]   # eight 4-plane slabs of the vertex class
out = {}
for seed in (0, 1, 2):
    emu.lib().emu_set_seed_study(seed)
    emu.set_fast(1)
    for b, e in runs:
        m.sample_range(dom, R, b, e)
    fs = emu.fast_stats()
    B = fs["bricks"]
    out[seed] = {k: fs[k] / B for k in ("pair_steps", "leaf_visits", "tri_pairs", "appends")}
    out[seed]["filter_pairs_step1"] = fs["hist"][16] / B
    print("%s %d^3  %s: %d bricks  pair steps %.2f  leaf visits %.2f  filter pairs step 1 %.2f  step 2 %.2f  appends %.2f"
          % (mesh, res, ("as the kernel   ", "oracle-tight seed", "centre + 2 radii ")[seed], B, out[seed]["pair_steps"], out[seed]["leaf_visits"],
             out[seed]["filter_pairs_step1"], out[seed]["tri_pairs"], out[seed]["appends"]))
emu.lib().emu_set_seed_study(0)
for k in out[0]:
    print("  %-20s oracle-tight %+.1f %%   centre + 2 brick radii %+.1f %%" % (k, 100.0 * (out[1][k] / out[0][k] - 1.0), 100.0 * (out[2][k] / out[0][k] - 1.0)))
