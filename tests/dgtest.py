"""Shared test tooling: deterministic meshes, OBJ reader, ctypes bindings of the two CPU
checkers (oracle restatement, and the unmodified reference when oracle/_ref is built).

TEST INFRASTRUCTURE -- nothing here is imported by the product package.
"""
import ctypes as C
import os
import re
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_ROOT = "/root/reference"

c_dp = C.POINTER(C.c_double)
c_up = C.POINTER(C.c_uint)
c_ip = C.POINTER(C.c_int)
c_u64p = C.POINTER(C.c_uint64)


def dp(a):
    return None if a is None else a.ctypes.data_as(c_dp)


def up(a):
    return None if a is None else a.ctypes.data_as(c_up)


def ip(a):
    return None if a is None else a.ctypes.data_as(c_ip)


# --------------------------------------------------------------------------------------
# meshes
# --------------------------------------------------------------------------------------
def load_obj(path):
    """OBJ subset the reference reads (discregrid/src/mesh/triangle_mesh.cpp:100-120):
    only 'v ' / 'f ' lines, index before the first '/', first three indices, 1-based."""
    V, F = [], []
    with open(path) as f:
        for line in f:
            if line[:2] == "v ":
                V.append([float(t) for t in line[2:].split()[:3]])
            elif line[:2] == "f ":
                F.append([int(t.split("/")[0]) - 1 for t in line[2:].split()[:3]])
    return np.array(V, dtype=np.float64), np.array(F, dtype=np.uint32)


def box_mesh():
    """The 8-vertex / 12-triangle cube of cmd/generate_sdf/resources/box.obj (values
    re-typed here; same vertex and face order so box.cdf is reproducible)."""
    V = np.array([[1, -1, -1], [1, -1, 1], [-1, -1, 1], [-1, -1, -1],
                  [1, 1, -1], [1, 1, 1], [-1, 1, 1], [-1, 1, -1]], dtype=np.float64)
    F = np.array([[2, 3, 4], [8, 7, 6], [5, 6, 2], [6, 7, 3], [3, 7, 8], [1, 4, 8],
                  [1, 2, 4], [5, 8, 6], [1, 5, 2], [2, 6, 3], [4, 3, 8], [5, 1, 8]], dtype=np.uint32) - 1
    return V, F


def icosphere(nu):
    """Class-I geodesic icosphere of frequency nu: 20 nu^2 triangles, 10 nu^2 + 2
    vertices on the unit sphere, outward orientation.  Fully deterministic (no RNG).
    nu = 71 -> 100 820 triangles (BASELINE.json config 3)."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    base = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t],
                     [0, -1, -t], [0, 1, -t], [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    base /= np.linalg.norm(base[0])
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2),
             (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5),
             (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    verts = {}
    out_v = []

    def vid(key, p):
        if key not in verts:
            verts[key] = len(out_v)
            out_v.append(p / np.linalg.norm(p))
        return verts[key]

    def key_of(a, b, c, i, j):
        # barycentric integer coordinates (i along a->b, j along a->c), canonical over
        # shared edges/corners
        k = nu - i - j
        w = {a: k, b: i, c: j}
        nz = tuple(sorted((v, n) for v, n in w.items() if n))
        return nz

    out_f = []
    for (a, b, c) in faces:
        A, B, Cc = base[a], base[b], base[c]
        idx = {}
        for i in range(nu + 1):
            for j in range(nu + 1 - i):
                p = (A * (nu - i - j) + B * i + Cc * j) / nu
                idx[(i, j)] = vid(key_of(a, b, c, i, j), p)
        for i in range(nu):
            for j in range(nu - i):
                out_f.append((idx[(i, j)], idx[(i + 1, j)], idx[(i, j + 1)]))
                if i + j < nu - 1:
                    out_f.append((idx[(i + 1, j)], idx[(i + 1, j + 1)], idx[(i, j + 1)]))
    return np.array(out_v, dtype=np.float64), np.array(out_f, dtype=np.uint32)


def torus(nu=24, nv=12, R=1.0, r=0.35):
    """Closed genus-1 mesh with concave regions (exercises edge/vertex pseudonormals)."""
    V = []
    for i in range(nu):
        for j in range(nv):
            a, b = 2 * np.pi * i / nu, 2 * np.pi * j / nv
            V.append([(R + r * np.cos(b)) * np.cos(a), (R + r * np.cos(b)) * np.sin(a), r * np.sin(b)])
    F = []
    for i in range(nu):
        for j in range(nv):
            p00 = i * nv + j
            p10 = ((i + 1) % nu) * nv + j
            p01 = i * nv + (j + 1) % nv
            p11 = ((i + 1) % nu) * nv + (j + 1) % nv
            F.append([p00, p10, p11])
            F.append([p00, p11, p01])
    return np.array(V, dtype=np.float64), np.array(F, dtype=np.uint32)


def bipyramid(valence=40, radius=1.0, height=0.6):
    """closed, outward-oriented bipyramid: a ring of `valence` vertices at z = 0 and two apexes of that valence -- the lanes of a brick
    next to an apex hold long candidate lists in the filtered K1 kernel (its pooled epilogue's test mesh)"""
    a = 2.0 * np.pi * np.arange(valence) / valence
    V = np.vstack([np.stack([radius * np.cos(a), radius * np.sin(a), np.zeros(valence)], axis=1), [[0.0, 0.0, height]], [[0.0, 0.0, -height]]])
    top, bot = valence, valence + 1
    F = [[i, (i + 1) % valence, top] for i in range(valence)] + [[(i + 1) % valence, i, bot] for i in range(valence)]
    return np.ascontiguousarray(V, dtype=np.float64), np.ascontiguousarray(F, dtype=np.uint32)


def bunny_mesh():
    """Stanford bunny staged as tests/golden/bunny.npz by tests/golden/make_golden.py
    (the OBJ itself only exists under /root/reference)."""
    z = np.load(os.path.join(GOLDEN, "bunny.npz"))
    return z["V"].astype(np.float64), z["F"].astype(np.uint32)


def dragon_mesh():
    """The reference's third sample mesh, cmd/generate_sdf/resources/dragon.obj (79 988 triangles), staged as
    tests/golden/dragon.npz by tests/golden/make_digests.py dragon128 (the OBJ itself only exists under /root/reference)."""
    z = np.load(os.path.join(GOLDEN, "dragon.npz"))
    return z["V"].astype(np.float64), z["F"].astype(np.uint32)


def write_obj(path, V, F):
    with open(path, "w") as f:
        for v in V:
            f.write("v %s %s %s\n" % tuple(repr(float(x)) for x in v))
        for t in F:
            f.write("f %d %d %d\n" % tuple(int(i) + 1 for i in t))


def read_cdf(path):
    """Parses the reference's .cdf/.cdm layout (cubic_lagrange_discrete_grid.cpp:678-719)."""
    b = open(path, "rb").read()
    o = 0

    def take(dtype, n):
        nonlocal o
        a = np.frombuffer(b, dtype=dtype, count=n, offset=o)
        o += a.nbytes
        return a

    out = {"domain": take(np.float64, 6).copy(), "res": take(np.uint32, 3).copy(), "cell": take(np.float64, 3).copy(),
           "inv_cell": take(np.float64, 3).copy(), "n_cells": int(take(np.uint64, 1)[0]),
           "n_fields": int(take(np.uint64, 1)[0])}
    nf = int(take(np.uint64, 1)[0])
    out["nodes"] = [take(np.float64, int(take(np.uint64, 1)[0])).copy() for _ in range(nf)]
    nf = int(take(np.uint64, 1)[0])
    out["cells"] = [take(np.uint32, 32 * int(take(np.uint64, 1)[0])).reshape(-1, 32).copy() for _ in range(nf)]
    nf = int(take(np.uint64, 1)[0])
    out["cell_map"] = [take(np.uint32, int(take(np.uint64, 1)[0])).copy() for _ in range(nf)]
    assert o == len(b), "trailing bytes in %s" % path
    return out


def n_nodes(res):
    nx, ny, nz = (int(r) for r in res)
    nv = (nx + 1) * (ny + 1) * (nz + 1)
    ne = nx * (ny + 1) * (nz + 1) + (nx + 1) * ny * (nz + 1) + (nx + 1) * (ny + 1) * nz
    return nv + 2 * ne


def rel_err(a, b):
    """max |a-b|/|b| over b != 0 (SURVEY.md section 8(c) parity procedure)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    m = b != 0
    if not m.any():
        return 0.0
    return float(np.max(np.abs(a[m] - b[m]) / np.abs(b[m])))


# --------------------------------------------------------------------------------------
# CPU checker #1: this repo's restatement (oracle/discregrid_oracle.cpp)
# --------------------------------------------------------------------------------------
_oracle = None


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])


def oracle_lib():
    global _oracle
    if _oracle is None:
        path = os.path.join(ORACLE_DIR, "libdiscregrid_oracle.so")
        src = os.path.join(ORACLE_DIR, "discregrid_oracle.cpp")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            build_oracle()
        L = C.CDLL(path)
        L.dgo_mesh_create.restype = C.c_void_p
        L.dgo_mesh_create.argtypes = [c_dp, C.c_size_t, c_up, C.c_size_t]
        L.dgo_mesh_free.argtypes = [C.c_void_p]
        L.dgo_mesh_watertight_flags.argtypes = [C.c_void_p]
        L.dgo_mesh_sizes.argtypes = [C.c_void_p] + [C.POINTER(C.c_size_t)] * 3
        L.dgo_mesh_get.argtypes = [C.c_void_p, c_dp, c_dp, c_dp, c_dp, c_ip]
        L.dgo_signed_distance.argtypes = [C.c_void_p, c_dp, C.c_size_t, c_dp, c_ip, c_ip, c_dp, c_u64p]
        L.dgo_point_triangle.argtypes = [c_dp, c_dp, c_dp, c_dp, C.c_size_t, c_dp, c_dp, c_ip, c_dp]
        L.dgo_default_domain.argtypes = [c_dp, C.c_size_t, c_dp]
        L.dgo_n_nodes.restype = C.c_uint
        L.dgo_n_nodes.argtypes = [c_up]
        L.dgo_grid_header.argtypes = [c_dp, c_up, c_dp, c_dp]
        L.dgo_node_positions.argtypes = [c_dp, c_up, C.c_uint, C.c_uint, c_dp]
        L.dgo_sample_nodes.restype = C.c_double
        L.dgo_sample_nodes.argtypes = [C.c_void_p, c_dp, c_up, C.c_int, C.c_uint, C.c_uint, c_dp, c_u64p]
        L.dgo_cell_table.argtypes = [c_up, C.c_uint, C.c_uint, c_up]
        L.dgo_shape_functions.argtypes = [c_dp, C.c_size_t, c_dp, c_dp]
        L.dgo_interpolate.restype = C.c_double
        L.dgo_interpolate.argtypes = [c_dp, c_up, c_dp, c_up, c_up, c_dp, C.c_size_t, c_dp, c_dp]
        L.dgo_density_map_nodes.restype = C.c_double
        L.dgo_density_map_nodes.argtypes = [c_dp, c_up, c_dp, c_up, c_up, C.c_double, C.c_double, C.c_int, c_dp, c_dp,
                                            C.c_uint, C.c_uint, c_dp]
        L.dgo_uniform_points.argtypes = [C.c_uint64, C.c_size_t, c_dp, c_dp, c_dp]
        L.dgo_write_cdf.restype = C.c_size_t
        L.dgo_write_cdf.argtypes = [C.c_char_p, c_dp, c_up, C.POINTER(c_dp), C.c_size_t]
        _oracle = L
    return _oracle


class OracleMesh:
    """TriangleMeshDistance restatement (BVH + pseudonormals + signed_distance)."""

    def __init__(self, V, F):
        self.L = oracle_lib()
        self.V = np.ascontiguousarray(V, dtype=np.float64)
        self.F = np.ascontiguousarray(F, dtype=np.uint32)
        self.h = self.L.dgo_mesh_create(dp(self.V), len(self.V), up(self.F), len(self.F))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.dgo_mesh_free(self.h)
            self.h = None

    def signed_distance(self, P, full=False, visits=False):
        P = np.ascontiguousarray(P, dtype=np.float64).reshape(-1, 3)
        n = len(P)
        d = np.empty(n)
        tri = np.empty(n, dtype=np.int32)
        ent = np.empty(n, dtype=np.int32)
        near = np.empty((n, 3))
        vis = np.zeros(2, dtype=np.uint64) if visits else None
        self.L.dgo_signed_distance(self.h, dp(P), n, dp(d), ip(tri), ip(ent), dp(near),
                                   None if vis is None else vis.ctypes.data_as(c_u64p))
        if visits:
            return d, vis
        return (d, tri, ent, near) if full else d

    def sample_nodes(self, domain, res, begin=0, end=None, invert=False, visits=False):
        domain = np.ascontiguousarray(domain, dtype=np.float64)
        res = np.ascontiguousarray(res, dtype=np.uint32)
        if end is None:
            end = n_nodes(res)
        out = np.empty(end - begin)
        vis = np.zeros(2, dtype=np.uint64) if visits else None
        secs = self.L.dgo_sample_nodes(self.h, dp(domain), up(res), int(invert), begin, end, dp(out),
                                       None if vis is None else vis.ctypes.data_as(c_u64p))
        self.last_seconds = secs
        return (out, vis) if visits else out

    def construction(self):
        nn, nt, nv = C.c_size_t(), C.c_size_t(), C.c_size_t()
        self.L.dgo_mesh_sizes(self.h, C.byref(nn), C.byref(nt), C.byref(nv))
        pt = np.empty((nt.value, 3))
        pe = np.empty((nt.value, 3, 3))
        pv = np.empty((nv.value, 3))
        sp = np.empty((nn.value, 8))
        ch = np.empty((nn.value, 2), dtype=np.int32)
        self.L.dgo_mesh_get(self.h, dp(pt), dp(pe), dp(pv), dp(sp), ip(ch))
        return dict(pn_tri=pt, pn_edge=pe, pn_vert=pv, spheres=sp, children=ch)


def oracle_point_triangle(V, F, tri, P):
    """Squared distance point <-> triangle with the reference's arithmetic (TriangleMeshDistance.h:564-820)
    for explicit pairs: tri and P may be one index / point (returns a float) or arrays (returns an array).
    Two triangles that are 'exactly tied' for a point give the same double here."""
    V = np.ascontiguousarray(V, dtype=np.float64)
    F = np.asarray(F)
    scalar = np.ndim(tri) == 0
    tri = np.atleast_1d(np.asarray(tri, dtype=np.int64))
    P = np.ascontiguousarray(np.asarray(P, dtype=np.float64).reshape(-1, 3))
    v0 = np.ascontiguousarray(V[F[tri, 0]])
    v1 = np.ascontiguousarray(V[F[tri, 1]])
    v2 = np.ascontiguousarray(V[F[tri, 2]])
    d2 = np.empty(len(tri))
    oracle_lib().dgo_point_triangle(dp(P), dp(v0), dp(v1), dp(v2), len(tri), dp(d2), None, None, None)
    return float(d2[0]) if scalar else d2


def assert_exact_ties(V, F, P, tri_a, tri_b):
    """Every pair of differing triangle ids names two triangles that are tied for the point under the
    reference's own acceptance rule.  The reference accepts a triangle only if its d^2 is below the SQUARE
    OF THE STORED sqrt of the best so far (TriangleMeshDistance.h:528-531), so two triangles are
    interchangeable exactly when their distances -- the correctly rounded sqrt of the reference's d^2 --
    are the same double (their d^2 may then differ in the last bit; which one is kept depends on the
    visiting order, and the stored distance is the same either way)."""
    diff = np.flatnonzero(np.asarray(tri_a) != np.asarray(tri_b))
    if len(diff) == 0:
        return 0
    P = np.asarray(P).reshape(-1, 3)
    a = oracle_point_triangle(V, F, np.asarray(tri_a)[diff], P[diff])
    b = oracle_point_triangle(V, F, np.asarray(tri_b)[diff], P[diff])
    bad = np.flatnonzero(np.sqrt(a) != np.sqrt(b))
    assert len(bad) == 0, "triangle ids differ without an exact tie at %s" % diff[bad][:8]
    return len(diff)


def oracle_default_domain(V):
    V = np.ascontiguousarray(V, dtype=np.float64)
    out = np.empty(6)
    oracle_lib().dgo_default_domain(dp(V), len(V), dp(out))
    return out


def oracle_node_positions(domain, res, begin=0, end=None):
    domain = np.ascontiguousarray(domain, dtype=np.float64)
    res = np.ascontiguousarray(res, dtype=np.uint32)
    if end is None:
        end = n_nodes(res)
    out = np.empty((end - begin, 3))
    oracle_lib().dgo_node_positions(dp(domain), up(res), begin, end, dp(out))
    return out


def oracle_cell_table(res, begin=0, end=None):
    res = np.ascontiguousarray(res, dtype=np.uint32)
    if end is None:
        end = int(res[0]) * int(res[1]) * int(res[2])
    out = np.empty((end - begin, 32), dtype=np.uint32)
    oracle_lib().dgo_cell_table(up(res), begin, end, up(out))
    return out


def oracle_shape(xi, grad=False):
    xi = np.ascontiguousarray(xi, dtype=np.float64).reshape(-1, 3)
    N = np.empty((len(xi), 32))
    dN = np.empty((len(xi), 32, 3)) if grad else None
    oracle_lib().dgo_shape_functions(dp(xi), len(xi), dp(N), dp(dN))
    return (N, dN) if grad else N


def oracle_interpolate(domain, res, coeffs, P, grad=False, cells=None, cell_map=None):
    domain = np.ascontiguousarray(domain, dtype=np.float64)
    res = np.ascontiguousarray(res, dtype=np.uint32)
    coeffs = np.ascontiguousarray(coeffs, dtype=np.float64)
    P = np.ascontiguousarray(P, dtype=np.float64).reshape(-1, 3)
    phi = np.empty(len(P))
    g = np.empty((len(P), 3)) if grad else None
    if cells is not None:
        cells = np.ascontiguousarray(cells, dtype=np.uint32)
    if cell_map is not None:
        cell_map = np.ascontiguousarray(cell_map, dtype=np.uint32)
    secs = oracle_lib().dgo_interpolate(dp(domain), up(res), dp(coeffs), up(cells), up(cell_map), dp(P), len(P),
                                        dp(phi), dp(g))
    oracle_interpolate.last_seconds = secs
    return (phi, g) if grad else phi


def uniform_points(seed, n, lo, hi):
    """n points uniform in [lo, hi] from std::mt19937_64(seed) (BASELINE config 5's query generator)."""
    lo = np.ascontiguousarray(lo, dtype=np.float64)
    hi = np.ascontiguousarray(hi, dtype=np.float64)
    out = np.empty((n, 3))
    oracle_lib().dgo_uniform_points(seed, n, dp(lo), dp(hi), dp(out))
    return out


def block_digests(a, block=1 << 20):
    """SHA-256 (first 16 bytes) of the raw bytes of every `block` consecutive items of a (rows of a 2-D
    array count as items): the committed lattice / query digests of tests/golden/make_digests.py."""
    import hashlib
    from concurrent.futures import ThreadPoolExecutor
    a = np.ascontiguousarray(a)
    n = a.shape[0]
    out = np.empty(((n + block - 1) // block, 16), dtype=np.uint8)

    def one(i):
        out[i] = np.frombuffer(hashlib.sha256(memoryview(a[i * block:(i + 1) * block]).cast("B")).digest()[:16], dtype=np.uint8)

    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:   # hashlib releases the GIL on large buffers
        list(ex.map(one, range(len(out))))
    return out


def gauss_rule_p30():
    """The reference's 16-point rule (table row p = 30 of gauss_quadrature.cpp, which lives in an
    unnamed namespace and cannot be linked): parsed from the reference source by make_golden.py and
    stored in the golden fixture."""
    z = np.load(os.path.join(GOLDEN, "ref_vectors.npz"))
    return z["gauss16_x"], z["gauss16_w"]


def parse_reference_gauss_rule(p=30):
    import re
    src = open(os.path.join(REF_ROOT, "cmd", "generate_density_map", "gauss_quadrature.cpp")).read()
    a = src.index("gaussian_abscissae_1[101][51]")
    b = src.index("gaussian_weights_1[101][51]")

    def row(s):
        i = s.index("// p = %d\n" % p)
        return np.array([float(v) for v in re.findall(r"[-+]?\d+\.\d+", s[i:s.index("}", i)])])
    return row(src[a:b]), row(src[b:])


def oracle_density_map(domain, res, coeffs, h, rho0, band=True, begin=0, end=None, cells=None, cell_map=None):
    domain = np.ascontiguousarray(domain, dtype=np.float64)
    res = np.ascontiguousarray(res, dtype=np.uint32)
    coeffs = np.ascontiguousarray(coeffs, dtype=np.float64)
    if end is None:
        end = n_nodes(res)
    gx, gw = gauss_rule_p30()
    gx, gw = np.ascontiguousarray(gx), np.ascontiguousarray(gw)
    out = np.empty(end - begin)
    if cells is not None:
        cells = np.ascontiguousarray(cells, dtype=np.uint32)
        cell_map = np.ascontiguousarray(cell_map, dtype=np.uint32)
    secs = oracle_lib().dgo_density_map_nodes(dp(domain), up(res), dp(coeffs), up(cells), up(cell_map), h, rho0,
                                              int(band), dp(gx), dp(gw), begin, end, dp(out))
    oracle_density_map.last_seconds = secs
    return out


def oracle_write_cdf(path, domain, res, fields):
    domain = np.ascontiguousarray(domain, dtype=np.float64)
    res = np.ascontiguousarray(res, dtype=np.uint32)
    fields = [np.ascontiguousarray(f, dtype=np.float64) for f in fields]
    arr = (c_dp * len(fields))(*[dp(f) for f in fields])
    return oracle_lib().dgo_write_cdf(path.encode(), dp(domain), up(res), arr, len(fields))


# --------------------------------------------------------------------------------------
# CPU checker #2: the unmodified reference (oracle/_ref), when built
# --------------------------------------------------------------------------------------
_ref = None


def ref_available():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libdiscregrid_ref.so"))


def ref_lib():
    global _ref
    if _ref is None:
        L = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libdiscregrid_ref.so"))
        L.ref_grid_create.restype = C.c_void_p
        L.ref_grid_create.argtypes = [c_dp, C.c_size_t, c_up, C.c_size_t, c_dp, c_up]
        L.ref_grid_load.restype = C.c_void_p
        L.ref_grid_load.argtypes = [C.c_char_p]
        L.ref_grid_free.argtypes = [C.c_void_p]
        L.ref_default_domain.argtypes = [c_dp, C.c_size_t, c_dp]
        L.ref_grid_add_sdf.restype = C.c_double
        L.ref_grid_add_sdf.argtypes = [C.c_void_p, C.c_int]
        L.ref_sample_nodes.restype = C.c_double
        L.ref_sample_nodes.argtypes = [C.c_void_p, C.c_int, C.c_uint, C.c_uint, c_dp]
        for f in ("ref_grid_n_fields",):
            getattr(L, f).restype = C.c_size_t
            getattr(L, f).argtypes = [C.c_void_p]
        for f in ("ref_grid_n_nodes", "ref_grid_n_cells"):
            getattr(L, f).restype = C.c_size_t
            getattr(L, f).argtypes = [C.c_void_p, C.c_uint]
        L.ref_grid_get_nodes.argtypes = [C.c_void_p, C.c_uint, c_dp]
        L.ref_grid_get_cells.argtypes = [C.c_void_p, C.c_uint, c_up]
        L.ref_grid_get_cell_map.argtypes = [C.c_void_p, C.c_uint, c_up]
        L.ref_grid_get_header.argtypes = [C.c_void_p, c_dp, c_up, c_dp, c_dp]
        L.ref_grid_add_coeffs.restype = C.c_uint
        L.ref_grid_add_coeffs.argtypes = [C.c_void_p, c_dp, C.c_size_t]
        L.ref_grid_save.argtypes = [C.c_void_p, C.c_char_p]
        L.ref_node_positions.argtypes = [C.c_void_p, C.c_uint, C.c_uint, c_dp]
        L.ref_grid_interpolate.restype = C.c_double
        L.ref_grid_interpolate.argtypes = [C.c_void_p, C.c_uint, c_dp, C.c_size_t, c_dp, c_dp]
        L.ref_grid_reduce_abs_lt.argtypes = [C.c_void_p, C.c_uint, C.c_double]
        L.ref_signed_distance.argtypes = [C.c_void_p, c_dp, C.c_size_t, c_dp, c_ip, c_ip, c_dp]
        L.ref_grid_add_density_map.restype = C.c_double
        L.ref_grid_add_density_map.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int]
        L.ref_grid_reduce_density.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.ref_density_nodes.restype = C.c_double
        L.ref_density_nodes.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_uint, C.c_uint, c_dp]
        L.ref_md_sizes.argtypes = [C.c_void_p] + [C.POINTER(C.c_size_t)] * 3
        L.ref_md_get.argtypes = [C.c_void_p, c_dp, c_dp, c_dp, c_dp, c_ip]
        _ref = L
    return _ref


def ref_default_domain(V):
    V = np.ascontiguousarray(V, dtype=np.float64)
    out = np.empty(6)
    ref_lib().ref_default_domain(dp(V), len(V), dp(out))
    return out


class RefGrid:
    """The reference's TriangleMesh + TriangleMeshDistance + CubicLagrangeDiscreteGrid."""

    def __init__(self, V=None, F=None, domain=None, res=None, path=None):
        self.L = ref_lib()
        if path is not None:
            self.h = self.L.ref_grid_load(path.encode())
            return
        domain = np.ascontiguousarray(domain, dtype=np.float64)
        res = np.ascontiguousarray(res, dtype=np.uint32)
        if V is not None:
            V = np.ascontiguousarray(V, dtype=np.float64)
            F = np.ascontiguousarray(F, dtype=np.uint32)
            self.h = self.L.ref_grid_create(dp(V), len(V), up(F), len(F), dp(domain), up(res))
        else:
            self.h = self.L.ref_grid_create(None, 0, None, 0, dp(domain), up(res))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.ref_grid_free(self.h)
            self.h = None

    def add_sdf(self, invert=False):
        return self.L.ref_grid_add_sdf(self.h, int(invert))

    def sample_nodes(self, begin, end, invert=False):
        out = np.empty(end - begin)
        self.last_seconds = self.L.ref_sample_nodes(self.h, int(invert), begin, end, dp(out))
        return out

    def add_coeffs(self, coeffs):
        coeffs = np.ascontiguousarray(coeffs, dtype=np.float64)
        return self.L.ref_grid_add_coeffs(self.h, dp(coeffs), len(coeffs))

    def nodes(self, f=0):
        out = np.empty(self.L.ref_grid_n_nodes(self.h, f))
        self.L.ref_grid_get_nodes(self.h, f, dp(out))
        return out

    def cells(self, f=0):
        out = np.empty((self.L.ref_grid_n_cells(self.h, f), 32), dtype=np.uint32)
        self.L.ref_grid_get_cells(self.h, f, up(out))
        return out

    def cell_map(self, f=0, n=None):
        hd = self.header()
        n = int(np.prod(hd["res"]))
        out = np.empty(n, dtype=np.uint32)
        self.L.ref_grid_get_cell_map(self.h, f, up(out))
        return out

    def header(self):
        dom = np.empty(6)
        res = np.empty(3, dtype=np.uint32)
        cell = np.empty(3)
        inv = np.empty(3)
        self.L.ref_grid_get_header(self.h, dp(dom), up(res), dp(cell), dp(inv))
        return dict(domain=dom, res=res, cell=cell, inv_cell=inv)

    def save(self, path):
        self.L.ref_grid_save(self.h, path.encode())

    def node_positions(self, begin, end):
        out = np.empty((end - begin, 3))
        self.L.ref_node_positions(self.h, begin, end, dp(out))
        return out

    def interpolate(self, P, f=0, grad=False):
        P = np.ascontiguousarray(P, dtype=np.float64).reshape(-1, 3)
        phi = np.empty(len(P))
        g = np.empty((len(P), 3)) if grad else None
        self.last_seconds = self.L.ref_grid_interpolate(self.h, f, dp(P), len(P), dp(phi), dp(g))
        return (phi, g) if grad else phi

    def add_density_map(self, h, rho0, no_reduction=False):
        return self.L.ref_grid_add_density_map(self.h, h, rho0, int(no_reduction))

    def density_nodes(self, h, rho0, begin, end, no_reduction=False):
        """The density-map tool's predicate + density_func over nodes [begin, end) of field 0 (no cell table)."""
        out = np.empty(end - begin)
        self.last_seconds = self.L.ref_density_nodes(self.h, h, rho0, int(no_reduction), begin, end, dp(out))
        return out

    def reduce_density(self, h, rho0):
        self.L.ref_grid_reduce_density(self.h, h, rho0)

    def reduce_abs_lt(self, f, bound):
        self.L.ref_grid_reduce_abs_lt(self.h, f, float(bound))

    def signed_distance(self, P, full=False):
        P = np.ascontiguousarray(P, dtype=np.float64).reshape(-1, 3)
        n = len(P)
        d = np.empty(n)
        tri = np.empty(n, dtype=np.int32)
        ent = np.empty(n, dtype=np.int32)
        near = np.empty((n, 3))
        self.L.ref_signed_distance(self.h, dp(P), n, dp(d), ip(tri), ip(ent), dp(near))
        return (d, tri, ent, near) if full else d

    def construction(self):
        nn, nt, nv = C.c_size_t(), C.c_size_t(), C.c_size_t()
        self.L.ref_md_sizes(self.h, C.byref(nn), C.byref(nt), C.byref(nv))
        pt = np.empty((nt.value, 3))
        pe = np.empty((nt.value, 3, 3))
        pv = np.empty((nv.value, 3))
        sp = np.empty((nn.value, 8))
        ch = np.empty((nn.value, 2), dtype=np.int32)
        self.L.ref_md_get(self.h, dp(pt), dp(pe), dp(pv), dp(sp), ip(ch))
        return dict(pn_tri=pt, pn_edge=pe, pn_vert=pv, spheres=sp, children=ch)


# ---- DG_FORCE: the library's test hooks and tuning knobs live behind ONE variable (discregrid_amd/csrc/dg_force.h) ----------
def _force_parse(txt):
    out = {}
    for item in re.split(r"[; \t]+", txt or ""):
        if "=" in item:
            k, v = item.split("=", 1)
            out[k] = v
    return out


def force_string(base=None, **kv):
    """DG_FORCE value: `base` (an existing DG_FORCE string) updated with key=value pairs; a value of None removes the key"""
    d = _force_parse(base)
    for k, v in kv.items():
        if v is None:
            d.pop(k, None)
        else:
            d[k] = str(v)
    return ";".join("%s=%s" % (k, v) for k, v in d.items())


def force(monkeypatch, **kv):
    """sets / removes keys of DG_FORCE in this process through pytest's monkeypatch (undone at teardown)"""
    v = force_string(os.environ.get("DG_FORCE"), **kv)
    if v:
        monkeypatch.setenv("DG_FORCE", v)
    else:
        monkeypatch.delenv("DG_FORCE", raising=False)


def force_env(env=None, **kv):
    """a copy of `env` (default: os.environ) with the given keys of DG_FORCE set, for subprocesses"""
    e = dict(os.environ if env is None else env)
    v = force_string(e.get("DG_FORCE"), **kv)
    if v:
        e["DG_FORCE"] = v
    else:
        e.pop("DG_FORCE", None)
    return e
