"""Diagnostics of one K1 launch on the GPU: time, heavy / redo bricks.  usage: python tools/k1_diag.py [nu] [res]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import dgtest as T
import discregrid_amd as dg

nu = int(sys.argv[1]) if len(sys.argv) > 1 else 71
res = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dg.load_library(); dg.set_device(0)
V, F = T.icosphere(nu)
dom = T.oracle_default_domain(V)
grid = dg.grid_desc(dom[:3], dom[3:], [res] * 3)
mesh = dg.Mesh(V, F)
n = dg.n_nodes(grid)
out = torch.empty(n, dtype=torch.float64, device="cuda")
for it in range(3):
    torch.cuda.synchronize(); t = time.time()
    mesh.sample_nodes_device(grid, 0, n, out.data_ptr())
    torch.cuda.synchronize(); dt = time.time() - t
print("leaf=%s fast=%s: %.3f ms, heavy (asked, split) = %s, info %s" % (
    os.environ.get("DG_FORCE", "-"), "-", dt * 1e3, mesh.last_heavy_bricks(),
    {k: v for k, v in mesh.info().items() if k in ("bvh_depth", "n_bvh_nodes")}))
