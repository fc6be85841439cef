import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import dgtest as T, discregrid_amd as dg
dg.load_library()
V, F = T.icosphere(71)
m = dg.Mesh(V, F)
rng = np.random.default_rng(0)
P = rng.uniform(-1.2, 1.2, size=(2000, 3))
m.signed_distance(P[:1])
t0 = time.perf_counter()
for i in range(2000):
    m.signed_distance(P[i:i + 1])
dt = time.perf_counter() - t0
print("single-point signed_distance through the host entry point: %.1f us per call" % (dt / 2000 * 1e6))
g = dg.grid_desc([-1.2] * 3, [1.2] * 3, [32] * 3)
f = dg.Field(g, m.sample_nodes(g))
f.interpolate(P[:1])
t0 = time.perf_counter()
for i in range(2000):
    f.interpolate(P[i:i + 1])
dt = time.perf_counter() - t0
print("single-query interpolate through the host entry point: %.1f us per call" % (dt / 2000 * 1e6))
