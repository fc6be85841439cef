"""Event rates of the filtered K1 traversal per brick (CPU emulator running the product's traversal template): what
tools/k1_opcode_hist.py weights the instruction blocks of k_sample_fast with.  Writes profiles/r05_k1_event_rates.json.
usage: python tests/perf/emu_event_rates.py [ico|bunny|dragon] [res]"""
import json
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import ctypes as C
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
runs = [(int(k) * 4 * plane, (int(k) * 4 + 4) * plane) for k in np.linspace(0, (res + 1) // 4 - 1, 8)]
ev = np.zeros(4, dtype=np.uint64)
emu.lib().emu_fast_events(ev.ctypes.data_as(C.c_void_p), 1)
emu.set_fast(1)
for b, e in runs:
    m.sample_range(dom, R, b, e)
fs = emu.fast_stats()
emu.lib().emu_fast_events(ev.ctypes.data_as(C.c_void_p), 1)
B = fs["bricks"]
out = {"mesh": mesh, "res": res, "bricks": B, "what": "eight 4-plane slabs of the vertex class, spread over the lattice",
       "per_brick": {"pair_steps": fs["pair_steps"] / B, "dead_steps": int(ev[0]) / B, "pushes": int(ev[1]) / B, "pops": int(ev[2]) / B,
                     "stale_pops": int(ev[3]) / B, "leaf_visits": fs["leaf_visits"] / B, "filter_pairs_step1": fs["hist"][16] / B,
                     "filter_pairs_step2": fs["tri_pairs"] / B, "lane_appends": fs["appends"] / B, "candidates_per_lane": fs["sum_list"] / max(1, fs["lanes"]),
                     "longest_list_per_wave": fs["sum_max_list"] / B, "bricks_with_exact_lanes": fs["redo_bricks"] / B, "parked": fs["parked"] / B}}
path = os.path.join(T.ROOT, "profiles", "r05_k1_event_rates_%s%d.json" % (mesh, res))
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out["per_brick"], indent=1))
